// Probe: the workgroup-row 2D kernel at TWELVE applications per launch -- four stages of three levels, one 16-wave
// workgroup per CU -- beside the six-application form (two stages, two 8-wave workgroups per CU): bit-identical?  how fast?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLORA_WG_K=6 -I../../include -I../../lorastencil_amd/csrc -c ../../lorastencil_amd/csrc/kernels_2d_wg.hip -o bin/wg_k6.o
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLORA_WG_K=12 -DLORA_WG_S=4 -I../../include -I../../lorastencil_amd/csrc -o bin/wg12_probe wg12_probe.hip bin/wg_k6.o
#include "kernels_2d_wg.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace lora {
hipError_t launch_2d_wg_k6(const Plan &p, ArgsWG a, int rows_total, hipStream_t s);
// the K = 6 object refers to the tails' entries: not used here
hipError_t launch_2d_wg_k4(const Plan &, ArgsWG, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_2d_wg_k2(const Plan &, ArgsWG, int, hipStream_t) { return hipErrorNotSupported; }
void set_last_error(const char *, hipError_t) {}
}  // namespace lora

static lora::ArgsWG args(const lora::Plan &p, int K, const double *in, double *out) {
    lora::ArgsWG a{};
    a.in = in;
    a.out = out;
    a.m = p.dims[0];
    a.n = p.dims[1];
    a.ld = a.n + 8;
    a.row_begin = 0;
    a.row_end = a.m;
    a.outw = 512 - 6 * K;
    a.strips = (a.n + a.outw - 1) / a.outw;
    return a;
}

int main(int argc, char **argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 16384, n = argc > 2 ? atoi(argv[2]) : 16384;
    lora::Plan p;
    p.ndim = 2;
    p.dims[0] = m;
    p.dims[1] = n;
    p.boundary = LORA_BC_REFERENCE;
    p.fused_eval = 7;  // nested profiles of the reference's star2d1r table, normalised
    const double tab[7][7] = {{0, 0, 0, 1, 0, 0, 0}, {0, 0, 2, 4, 2, 0, 0}, {0, 2, 4, 8, 4, 2, 0}, {1, 4, 8, 16, 8, 4, 1},
                              {0, 2, 4, 8, 4, 2, 0}, {0, 0, 2, 4, 2, 0, 0}, {0, 0, 0, 1, 0, 0, 0}};
    for (int i = 0; i < 49; ++i) p.w[i] = tab[i / 7][i % 7] / 100.0;
    p.nest_g[0] = 0.01, p.nest_g[1] = 0.02, p.nest_g[2] = 0.02, p.nest_g[3] = 0.01;
    p.nest_a[0] = 0, p.nest_a[1] = 2, p.nest_a[2] = 2, p.nest_a[3] = 4;
    const size_t count = (size_t) (m + 8) * (n + 8);
    double *b[3];
    for (auto &x : b)
        if (hipMalloc(&x, count * 8) != hipSuccess) return 1;
    std::vector<double> h(count);
    for (size_t i = 0; i < count; ++i) h[i] = (double) ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (auto &x : b) hipMemcpy(x, h.data(), count * 8, hipMemcpyHostToDevice);  // every buffer carries the same halo
    // parity: 12 applications in one launch == two launches of six
    lora::launch_2d_wg_k6(p, args(p, 6, b[0], b[1]), m, nullptr);
    lora::launch_2d_wg_k6(p, args(p, 6, b[1], b[2]), m, nullptr);
    if (lora::launch_2d_wg_k12(p, args(p, 12, b[0], b[1]), m, nullptr) != hipSuccess) {
        printf("K = 12 launch failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    hipDeviceSynchronize();
    std::vector<double> r6(count), r12(count);
    hipMemcpy(r6.data(), b[2], count * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r12.data(), b[1], count * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 4; i < m + 4; ++i)
        for (int j = 4; j < n + 4; ++j)
            if (memcmp(&r6[(size_t) i * (n + 8) + j], &r12[(size_t) i * (n + 8) + j], 8)) {
                if (bad < 5) printf("differs at (%d, %d): %.17g vs %.17g\n", i - 4, j - 4, r6[(size_t) i * (n + 8) + j], r12[(size_t) i * (n + 8) + j]);
                ++bad;
            }
    printf("%d x %d: K = 12 vs 2 x K = 6: %zu of %zu interior points differ\n", m, n, bad, (size_t) m * n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int K : {6, 12, 6, 12}) {
        const int it = 10;
        for (int w = 0; w < 2; ++w)
            K == 6 ? lora::launch_2d_wg_k6(p, args(p, 6, b[0], b[1]), m, nullptr) : lora::launch_2d_wg_k12(p, args(p, 12, b[0], b[1]), m, nullptr);
        hipEventRecord(e0);
        for (int i = 0; i < it; ++i) {
            const double *src = b[i & 1];
            double *dst = b[1 - (i & 1)];
            K == 6 ? lora::launch_2d_wg_k6(p, args(p, 6, src, dst), m, nullptr) : lora::launch_2d_wg_k12(p, args(p, 12, src, dst), m, nullptr);
        }
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / it;
        printf("K = %2d: %8.1f us per launch, %7.1f GStencils/s\n", K, us, (double) m * n * K / us / 1e3);
    }
    return 0;
}
