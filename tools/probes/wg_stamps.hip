// Diagnostic build of the workgroup-row 2D kernel (kernels_2d_wg.hip compiled with -DLORA_DIAGNOSTICS): every workgroup
// stamps s_memrealtime at entry and exit plus its HW_ID / XCC_ID, so that the launch's timeline -- who starts when, on
// which CU, who sets the pace -- can be read.  star2d1r taps, 16384^2 by default.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLORA_DIAGNOSTICS -I../../include -I../../lorastencil_amd/csrc \
//         -o bin/wg_stamps wg_stamps.hip && ./bin/wg_stamps [m n [wg_rows [edge_pct [prio]]]] > stamps.csv
#define LORA_WG_ONLY_K6
#include "kernels_2d_wg.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace lora {
// the two host helpers the launcher's Plan drags in are not needed here
}

int main(int argc, char **argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 16384, n = argc > 2 ? atoi(argv[2]) : 16384;
    lora::Plan p;
    p.ndim = 2;
    p.dims[0] = m;
    p.dims[1] = n;
    p.wg_rows = argc > 3 ? atoi(argv[3]) : 0;
    p.wg_edge_pct = argc > 4 ? atoi(argv[4]) : -1;
    if (argc > 5) p.wg_prio = atoi(argv[5]);
    // star2d1r, normalised: nested-profile form g = (1, 2, 2, 1) / 256, a = (., 2, 2, 4)   (2d/main.cu:187-195)
    p.fused_eval = lora::EVAL_NEST;
    const double g[4] = {1.0 / 256, 2.0 / 256, 2.0 / 256, 1.0 / 256}, a[4] = {0, 2, 2, 4};
    for (int k = 0; k < 4; ++k) {
        p.nest_g[k] = g[k];
        p.nest_a[k] = a[k];
    }
    const size_t count = (size_t) (m + 8) * (n + 8);
    double *b0, *b1;
    if (hipMalloc(&b0, count * 8) != hipSuccess || hipMalloc(&b1, count * 8) != hipSuccess) return 1;
    std::vector<double> h(count);
    for (size_t i = 0; i < count; ++i) h[i] = (double) ((i * 2654435761u) % 1000) / 1000.0;
    hipMemcpy(b0, h.data(), count * 8, hipMemcpyHostToDevice);
    hipMemset(b1, 0, count * 8);
    const int maxwg = 8192;
    long long *st;
    hipMalloc(&st, sizeof(long long) * 4 * maxwg);
    hipMemset(st, 0, sizeof(long long) * 4 * maxwg);
    for (int i = 0; i < 6; ++i) {  // warm up without stamps
        lora::launch_2d_wg(p, 6, b0, b1, 0, m, nullptr);
        lora::launch_2d_wg(p, 6, b1, b0, 0, m, nullptr);
    }
    hipDeviceSynchronize();
    lora::g_wg_stamps = st;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    lora::launch_2d_wg(p, 6, b0, b1, 0, m, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> s(4 * maxwg);
    hipMemcpy(s.data(), st, sizeof(long long) * 4 * maxwg, hipMemcpyDeviceToHost);
    long long t0 = -1;
    for (int b = 0; b < maxwg; ++b)
        if (s[4 * b] && (t0 < 0 || s[4 * b] < t0)) t0 = s[4 * b];
    printf("# launch %.1f us; columns: block,start_us,end_us,se,cu,xcc,strip,chunk\n", ms * 1e3);
    for (int b = 0; b < maxwg; ++b) {
        if (!s[4 * b]) continue;
        const unsigned hw = (unsigned) s[4 * b + 2], xcc = (unsigned) (s[4 * b + 2] >> 32) & 15;
        printf("%d,%.2f,%.2f,%u,%u,%u,%d,%d\n", b, (s[4 * b] - t0) / 100.0, (s[4 * b + 1] - t0) / 100.0, (hw >> 13) & 7, (hw >> 8) & 15,
               xcc, (int) (s[4 * b + 3] >> 32), (int) (s[4 * b + 3] & 0xffffffff));
    }
    return 0;
}
