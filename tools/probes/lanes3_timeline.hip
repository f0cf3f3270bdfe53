// When does each workgroup of a launch of the register-resident fp64 3D kernel start and end, and how long do rim tiles
// (EDGE steps throughout) take against inner ones?  kernels_3d_lanes.hip with -DLORA_L3_STAMP: per workgroup two readings
// of the 100 MHz clock, its tile and first plane.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLORA_L3_STAMP -I../../include -I../../lorastencil_amd/csrc -o bin/lanes3_timeline lanes3_timeline.hip
//   bin/lanes3_timeline [star|box] [d] [spans3]
namespace lora {
namespace {
long long *g_l3_stamps = nullptr;
long g_l3_stamp_blocks = 0;
}  // namespace
}  // namespace lora
#include "kernels_3d_lanes.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    const bool sep = argc > 1 && !strcmp(argv[1], "box");
    const int d = argc > 2 ? atoi(argv[2]) : (sep ? 768 : 512);
    lora::Plan p;
    p.ndim = 3;
    p.dims[0] = p.dims[1] = p.dims[2] = d;
    p.dtype = LORA_F64;
    p.boundary = LORA_BC_REFERENCE;
    p.tapset = sep ? lora::TAPS3D_BOX : lora::TAPS3D_STAR;
    p.sep64_valid = sep;
    p.spans3 = argc > 3 ? atoi(argv[3]) : -1;
    for (int k = 0; k < 27; ++k) p.w[k] = 0.0;
    p.w[4] = p.w[10] = p.w[12] = p.w[14] = p.w[16] = p.w[22] = 0.125;
    p.w[13] = 0.25;
    const double c[3] = {0.25, 0.5, 0.25};
    for (int k = 0; k < 9; ++k) p.sep64[k] = c[k % 3];
    const size_t count = (size_t) (d + 2) * (d + 4) * (d + 8);
    double *b0, *b1;
    if (hipMalloc(&b0, count * 8) != hipSuccess || hipMalloc(&b1, count * 8) != hipSuccess) return 1;
    std::vector<double> h(count);
    for (size_t i = 0; i < count; ++i) h[i] = (double) ((i * 2654435761u) % 1000) / 1000.0;
    hipMemcpy(b0, h.data(), count * 8, hipMemcpyHostToDevice);
    hipMemset(b1, 0, count * 8);
    const long cap = 1 << 16;
    hipMalloc(&lora::g_l3_stamps, cap * 4 * sizeof(long long));
    for (int i = 0; i < 6; ++i) {  // warm
        lora::launch_3d_lanes(p, 4, b0, b1, 0, d, nullptr);
        lora::launch_3d_lanes(p, 4, b1, b0, 0, d, nullptr);
    }
    hipDeviceSynchronize();
    hipMemset(lora::g_l3_stamps, 0, cap * 4 * sizeof(long long));
    lora::launch_3d_lanes(p, 4, b0, b1, 0, d, nullptr);
    hipDeviceSynchronize();
    const long n = lora::g_l3_stamp_blocks;
    std::vector<long long> st(n * 4);
    hipMemcpy(st.data(), lora::g_l3_stamps, n * 4 * sizeof(long long), hipMemcpyDeviceToHost);
    long long t0 = st[0], t1 = st[1];
    for (long b = 0; b < n; ++b) {
        t0 = std::min(t0, st[4 * b]);
        t1 = std::max(t1, st[4 * b + 1]);
    }
    // durations by class, starts by rank of start time
    std::vector<double> rim, inner, starts, ends;
    for (long b = 0; b < n; ++b) {
        const double us = (double) (st[4 * b + 1] - st[4 * b]) / 100.0;
        ((st[4 * b + 2] >> 31) & 1 ? rim : inner).push_back(us);
        starts.push_back((double) (st[4 * b] - t0) / 100.0);
        ends.push_back((double) (st[4 * b + 1] - t0) / 100.0);
    }
    auto stat = [](std::vector<double> v, const char *name) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        printf("  %-6s %5zu workgroups (last segment): duration min %.1f  median %.1f  p90 %.1f  max %.1f us\n", name, v.size(), v.front(),
               v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    printf("%s %d^3, spans3 = %d: %ld workgroups, launch %.1f us from first start to last end\n", sep ? "sep-box" : "star", d, p.spans3, n,
           (double) (t1 - t0) / 100.0);
    stat(rim, "rim");
    stat(inner, "inner");
    std::sort(starts.begin(), starts.end());
    std::sort(ends.begin(), ends.end());
    printf("  starts: 256th %.1f us, median %.1f, last %.1f;  ends: first %.1f, median %.1f, 90 %% %.1f, last %.1f\n",
           starts[std::min<long>(255, n - 1)], starts[n / 2], starts.back(), ends.front(), ends[n / 2], ends[n * 9 / 10], ends.back());
    // how many workgroups are running at 10 points in time
    printf("  running at 10 %% .. 100 %% of the launch:");
    for (int q = 1; q <= 10; ++q) {
        const double t = (double) (t1 - t0) / 100.0 * q / 10.0 - 0.01;
        long run = 0;
        for (long b = 0; b < n; ++b) run += ((double) (st[4 * b] - t0) / 100.0 <= t && (double) (st[4 * b + 1] - t0) / 100.0 > t);
        printf(" %ld", run);
    }
    printf("\n");
    // the slowest workgroups, and the mean duration per tile column / tile row
    std::vector<long> order(n);
    for (long b = 0; b < n; ++b) order[b] = b;
    std::sort(order.begin(), order.end(), [&](long x, long y) { return st[4 * x + 1] - st[4 * x] > st[4 * y + 1] - st[4 * y]; });
    printf("  slowest:");
    for (long i = 0; i < std::min<long>(n, 16); ++i) {
        const long b = order[i];
        printf(" (tx %lld ty %lld z %lld: %.0f)", st[4 * b + 2] & 0xffff, (st[4 * b + 2] >> 16) & 0x7fff, st[4 * b + 3], (double) (st[4 * b + 1] - st[4 * b]) / 100.0);
    }
    printf("\n");
    double sx[64] = {0}, sy[64] = {0};
    long nx[64] = {0}, ny[64] = {0};
    for (long b = 0; b < n; ++b) {
        const int tx = (int) (st[4 * b + 2] & 0xffff), ty = (int) ((st[4 * b + 2] >> 16) & 0x7fff);
        const double us = (double) (st[4 * b + 1] - st[4 * b]) / 100.0;
        if (tx < 64) sx[tx] += us, nx[tx] += 1;
        if (ty < 64) sy[ty] += us, ny[ty] += 1;
    }
    printf("  mean by tile column:");
    for (int i = 0; i < 64 && nx[i]; ++i) printf(" %.0f", sx[i] / nx[i]);
    printf("\n  mean by tile row:");
    for (int i = 0; i < 64 && ny[i]; ++i) printf(" %.0f", sy[i] / ny[i]);
    printf("\n");
    return 0;
}
