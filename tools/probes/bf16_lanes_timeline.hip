// Start / end of every workgroup of one launch of the register-resident bf16 kernel (four sweeps), the steps it ran and the
// segments it ran them in: is the launch even?  kernels_3d_bf16_lanes.hip with -DLORA_BL_TIMELINE.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -DLORA_BL_TIMELINE -I../../include -I../../lorastencil_amd/csrc \
//         -o bin/bf16_lanes_timeline bf16_lanes_timeline.hip && bin/bf16_lanes_timeline [h m n [spans3]]
namespace lora {
namespace {
long long *g_bl_timeline = nullptr;
long g_bl_timeline_blocks = 0;
}  // namespace
}  // namespace lora
#include "kernels_3d_bf16_lanes.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    const int h = argc > 3 ? atoi(argv[1]) : 768, m = argc > 3 ? atoi(argv[2]) : 768, n = argc > 3 ? atoi(argv[3]) : 768;
    lora::Plan p;
    p.ndim = 3;
    p.dims[0] = h;
    p.dims[1] = m;
    p.dims[2] = n;
    p.dtype = LORA_BF16;
    p.boundary = LORA_BC_REFERENCE;
    p.tapset = lora::TAPS3D_SEP;
    p.spans3 = argc > 4 ? atoi(argv[4]) : -1;
    const float c[3] = {0.25f, 0.5f, 0.25f};
    for (int k = 0; k < 9; ++k) p.sep[k] = c[k % 3];
    const size_t count = (size_t) (h + 2) * (m + 4) * (n + 8);
    unsigned short *b0, *b1;
    if (hipMalloc(&b0, count * 2) != hipSuccess || hipMalloc(&b1, count * 2) != hipSuccess) return 1;
    std::vector<unsigned short> hb(count);
    for (size_t i = 0; i < count; ++i) hb[i] = (unsigned short) (0x3f00 + (i * 2654435761u) % 128);
    (void) hipMemcpy(b0, hb.data(), count * 2, hipMemcpyHostToDevice);
    (void) hipMemset(b1, 0, count * 2);
    const long cap = 1 << 16;
    (void) hipMalloc(&lora::g_bl_timeline, cap * 4 * sizeof(long long));
    for (int i = 0; i < 6; ++i) {
        (void) lora::launch_3d_bf16_lanes(p, 4, b0, b1, 0, h, nullptr);
        (void) lora::launch_3d_bf16_lanes(p, 4, b1, b0, 0, h, nullptr);
    }
    (void) hipDeviceSynchronize();
    (void) hipMemset(lora::g_bl_timeline, 0, cap * 4 * sizeof(long long));
    (void) lora::launch_3d_bf16_lanes(p, 4, b0, b1, 0, h, nullptr);
    (void) hipDeviceSynchronize();
    const long nb = lora::g_bl_timeline_blocks;
    std::vector<long long> st(nb * 4);
    (void) hipMemcpy(st.data(), lora::g_bl_timeline, nb * 4 * sizeof(long long), hipMemcpyDeviceToHost);
    long long t0 = st[0], t1 = st[1];
    for (long b = 0; b < nb; ++b) {
        t0 = std::min(t0, st[4 * b]);
        t1 = std::max(t1, st[4 * b + 1]);
    }
    std::vector<double> dur, per_step, starts, ends;
    long steps_min = 1 << 30, steps_max = 0, segs_max = 0;
    for (long b = 0; b < nb; ++b) {
        const double us = (double) (st[4 * b + 1] - st[4 * b]) / 100.0;
        dur.push_back(us);
        per_step.push_back(us / (double) std::max<long long>(st[4 * b + 2], 1));
        starts.push_back((double) (st[4 * b] - t0) / 100.0);
        ends.push_back((double) (st[4 * b + 1] - t0) / 100.0);
        steps_min = std::min<long>(steps_min, (long) st[4 * b + 2]);
        steps_max = std::max<long>(steps_max, (long) st[4 * b + 2]);
        segs_max = std::max<long>(segs_max, (long) st[4 * b + 3]);
    }
    auto q = [](std::vector<double> v, double f) {
        std::sort(v.begin(), v.end());
        return v[std::min<size_t>(v.size() - 1, (size_t) (f * (double) v.size()))];
    };
    printf("bf16 sep-box %d x %d x %d, spans3 = %d: %ld workgroups, launch %.1f us first start to last end\n", h, m, n, p.spans3, nb,
           (double) (t1 - t0) / 100.0);
    printf("  steps per workgroup %ld .. %ld, at most %ld segments;  duration min %.1f  median %.1f  p90 %.1f  max %.1f us;  us per step median %.3f  p90 %.3f  max %.3f\n",
           steps_min, steps_max, segs_max, q(dur, 0.0), q(dur, 0.5), q(dur, 0.9), q(dur, 1.0), q(per_step, 0.5), q(per_step, 0.9), q(per_step, 1.0));
    printf("  starts: median %.1f, last %.1f;  ends: first %.1f, median %.1f, last %.1f\n", q(starts, 0.5), q(starts, 1.0), q(ends, 0.0), q(ends, 0.5),
           q(ends, 1.0));
    printf("  running at 10 %% .. 100 %% of the launch:");
    for (int k = 1; k <= 10; ++k) {
        const double t = (double) (t1 - t0) / 100.0 * k / 10.0 - 0.01;
        long run = 0;
        for (long b = 0; b < nb; ++b) run += (starts[b] <= t && ends[b] > t);
        printf(" %ld", run);
    }
    printf("\n");
    return 0;
}
