// Where does a step of the register-resident bf16 kernel spend its time?  kernels_3d_bf16_lanes.hip compiled with one part
// of the step removed at a time (-DLORA_BL_ABLATE=bits: 1 barriers, 2 stores, 4 plane loads, 8 no EDGE steps, 16 only EDGE
// steps; 32 = scalar taps, the only one of them whose results are right), timed on the separable box 768^3.
//   for b in 0 32 1 2 4 6 7 8 16; do hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -DLORA_BL_ABLATE=$b \
//       -I../../include -I../../lorastencil_amd/csrc -o bin/bf16_lanes_ablate_$b bf16_lanes_ablate.hip; done
#ifdef LORA_BL_STAMP
namespace lora {
namespace {
long long *g_bl_stamps = nullptr;
long g_bl_stamp_blocks = 0;
}  // namespace
}  // namespace lora
#endif
#include "kernels_3d_bf16_lanes.hip"

#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <vector>

static void run(int h, int m, int n, int zc) {
    lora::Plan p;
    p.ndim = 3;
    p.dims[0] = h;
    p.dims[1] = m;
    p.dims[2] = n;
    p.dtype = LORA_BF16;
    p.boundary = LORA_BC_REFERENCE;
    p.tapset = lora::TAPS3D_SEP;
    p.fused_z_chunk = zc;
    const float c[3] = {0.25f, 0.5f, 0.25f};
    for (int k = 0; k < 9; ++k) p.sep[k] = c[k % 3];
    const size_t count = (size_t) (h + 2) * (m + 4) * (n + 8);
    unsigned short *b0, *b1;
    if (hipMalloc(&b0, count * 2) != hipSuccess || hipMalloc(&b1, count * 2) != hipSuccess) exit(1);
    std::vector<unsigned short> hb(count);
    const bool wide = getenv("LORA_BL_WIDE_DATA") != nullptr;  // values of both signs over many binades instead of 0.5 .. 1.0
    for (size_t i = 0; i < count; ++i) {
        const unsigned r = (unsigned) (i * 2654435761u) ^ (unsigned) (i >> 7) * 40503u;
        hb[i] = wide ? (unsigned short) (((r >> 3) & 0x8000) | (0x3a00 + (r % 0x600))) : (unsigned short) (0x3f00 + r % 0x100);
    }
    hipMemcpy(b0, hb.data(), count * 2, hipMemcpyHostToDevice);
    hipMemset(b1, 0, count * 2);
#if LORA_BL_STAMP
    hipMalloc(&lora::g_bl_stamps, 8 * 2 * 16 * 4096);
    hipMemset(lora::g_bl_stamps, 0, 8 * 2 * 16 * 4096);
#endif
    for (int i = 0; i < 3; ++i) {
        lora::launch_3d_bf16_lanes(p, 4, b0, b1, 0, h, nullptr);
        lora::launch_3d_bf16_lanes(p, 4, b1, b0, 0, h, nullptr);
    }
    hipDeviceSynchronize();
#if LORA_BL_STAMP
    {
        const long nb = lora::g_bl_stamp_blocks;
        std::vector<long long> st(nb * 16 * 2);
        hipMemcpy(st.data(), lora::g_bl_stamps, st.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> inner, rim;
        for (long i = 0; i < nb * 16; ++i) {
            const long long n = st[2 * i + 1] & 0xffffffffLL;
            if (n > 0) (((st[2 * i + 1] >> 32) & 1) ? rim : inner).push_back((double) st[2 * i] / (double) n);
        }
        std::sort(inner.begin(), inner.end());
        std::sort(rim.begin(), rim.end());
        printf("phase %d: cycles per step and wave, median over %zu waves of interior tiles %.0f (10 %% %.0f, 90 %% %.0f); %zu waves of rim tiles %.0f\n",
               LORA_BL_STAMP, inner.size(), inner.empty() ? 0.0 : inner[inner.size() / 2], inner.empty() ? 0.0 : inner[inner.size() / 10],
               inner.empty() ? 0.0 : inner[inner.size() * 9 / 10], rim.size(), rim.empty() ? 0.0 : rim[rim.size() / 2]);
    }
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it = 10;
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) {
        lora::launch_3d_bf16_lanes(p, 4, b0, b1, 0, h, nullptr);
        lora::launch_3d_bf16_lanes(p, 4, b1, b0, 0, h, nullptr);
    }
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (2 * it);
    printf("ablate=%2d sep-box bf16 %d x %d x %d zc=%d: %8.1f us per launch, %7.1f GStencils/s\n", LORA_BL_ABLATE, h, m, n, zc, us,
           (double) h * m * n * 4 / us / 1e3);
    hipFree(b0);
    hipFree(b1);
}

int main(int argc, char **argv) {
    if (argc > 1) {  // 768^3 at the given z-chunk lengths
        for (int i = 1; i < argc; ++i) run(768, 768, 768, atoi(argv[i]));
        return 0;
    }
    run(768, 768, 768, 0);
    run(96, 768, 768, 0);
    return 0;
}
