// Probe: issue cost of v_mov_b32_dpp wave_shr:1 / wave_shl:1 (cross-lane neighbour access without LDS) beside v_fma_f64 on
// gfx950 -- the price of taking a radius-1 stencil's x-neighbours from the neighbouring lane's registers.
//   hipcc --offload-arch=gfx950 -O2 -o bin/dpp_rate_probe dpp_rate_probe.hip && ./bin/dpp_rate_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int ITER = 2000;

template <int NDPP, int NFMA, int NF32>
__global__ __launch_bounds__(256) void probe(long long *cycles, double *sink, double a, double b) {
    double x[8];
    int m[8];
    float f[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = threadIdx.x * 1e-9 + i;
        m[i] = threadIdx.x + i;
        f[i] = threadIdx.x * 1e-3f + i;
    }
    float fa = (float) a, fb = (float) b;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int v = 0; v < NDPP; ++v)
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(m[v & 7]) : "v"(m[(v + 1) & 7]));
#pragma unroll
        for (int v = 0; v < NFMA; ++v) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[v & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int v = 0; v < NF32; ++v)
            asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(f[v & 7]) : "v"(fa), "v"(fb));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + m[i] + f[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NDPP, int NFMA, int NF32>
void run(const char *label, int blocks_per_cu) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * blocks_per_cu, waves = blocks * 4;
    long long *dc;
    double *ds;
    hipMalloc(&dc, sizeof(long long) * waves);
    hipMalloc(&ds, 8);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<NDPP, NFMA, NF32>), dim3(blocks), dim3(256), 0, 0, dc, ds, 1.0000001, 0.9999999);
    hipDeviceSynchronize();
    std::vector<long long> h(waves);
    hipMemcpy(h.data(), dc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s waves/SIMD %d  dpp %2d  fma_f64 %2d  fmac_f32_dpp %2d  cycles/iter %8.2f\n", label, blocks_per_cu, NDPP, NFMA, NF32,
           (double) h[waves / 2] / ITER);
    hipFree(dc);
    hipFree(ds);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<16, 0, 0>("v_mov_b32_dpp wave_shr only", w);
        run<0, 16, 0>("v_fma_f64 only", w);
        run<16, 16, 0>("both", w);
        run<4, 16, 0>("4 dpp movs per 16 fma", w);
        run<0, 0, 16>("v_fmac_f32 with dpp operand", w);
    }
    return 0;
}
