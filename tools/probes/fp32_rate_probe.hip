// Probe: what the 32-bit vector instructions of a bf16 / fp32 stencil level cost on gfx950, one to four waves per SIMD:
// plain and packed fp32 multiply-adds, the DPP forms that fold a neighbour-lane read into the arithmetic, the bf16
// unpack (shift / and) and pack (v_cvt_pk_bf16_f32) and the bit-field insert an EDGE blend needs.  The question behind it:
// does v_pk_fma_f32 do two points for the price of one (then a lane should own aligned column pairs and keep DPP operands
// rare), or is it twice the pipe time of v_fma_f32 (then packing only saves issue slots)?
//   hipcc --offload-arch=gfx950 -O2 -o bin/fp32_rate_probe fp32_rate_probe.hip && ./bin/fp32_rate_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int ITER = 2000;
constexpr int N = 16;  // instructions per iteration, over 8 independent destination registers

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { FMA32, FMAC32, PKFMA32, PKMUL32, FMAC_DPP_WAVE, FMAC_DPP_ROW, MUL_DPP_WAVE, MOV_DPP_WAVE, LSHL, AND, CVTPK, BFI, FMA64, PKFMA_AND_FMA, PKADD32 };

template <int OP>
__global__ __launch_bounds__(256) void probe(long long *cycles, float *sink, float a, float b) {
    float f[8];
    f2 p[8];
    unsigned u[8];
    double d[8];
    for (int i = 0; i < 8; ++i) {
        f[i] = threadIdx.x * 1e-3f + i;
        p[i] = (f2){f[i], f[i] + 0.5f};
        u[i] = threadIdx.x * 2654435761u + i;
        d[i] = f[i];
    }
    const f2 a2 = {a, a}, b2 = {b, b};
    const double da = a, db = b;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int v = 0; v < N; ++v) {
            const int k = v & 7, k1 = (v + 1) & 7;
            if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "v"(a), "v"(b));
            if constexpr (OP == FMAC32) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "v"(a), "v"(b));
            if constexpr (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(a2), "v"(b2));
            if constexpr (OP == PKMUL32) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[k]) : "v"(a2), "v"(p[k1]));
            if constexpr (OP == PKADD32) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[k]) : "v"(a2), "v"(p[k1]));
            if constexpr (OP == FMAC_DPP_WAVE)
                asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(f[k]) : "v"(a), "v"(b));
            if constexpr (OP == FMAC_DPP_ROW)
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(f[k]) : "v"(a), "v"(b));
            if constexpr (OP == MUL_DPP_WAVE)
                asm volatile("v_mul_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f[k]) : "v"(f[k1]), "v"(b));
            if constexpr (OP == MOV_DPP_WAVE)
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[k]) : "v"(u[k1]));
            if constexpr (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[k]) : "v"(u[k1]));
            if constexpr (OP == AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[k]) : "v"(u[k1]));
            if constexpr (OP == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[k]) : "v"(f[k]), "v"(f[k1]));
            if constexpr (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(u[k]) : "v"(u[k1]), "v"(u[(v + 2) & 7]), "v"(u[(v + 3) & 7]));
            if constexpr (OP == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(da), "v"(db));
            if constexpr (OP == PKFMA_AND_FMA) {
                if (v & 1)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(a2), "v"(b2));
                else
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "v"(a), "v"(b));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + p[i].x + p[i].y + (float) u[i] + (float) d[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *label) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%-34s", label);
    for (int per_cu : {1, 2, 4}) {
        const int blocks = prop.multiProcessorCount * per_cu, waves = blocks * 4;
        long long *dc;
        float *ds;
        hipMalloc(&dc, sizeof(long long) * waves);
        hipMalloc(&ds, 8);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<OP>), dim3(blocks), dim3(256), 0, 0, dc, ds, 1.0000001f, 0.9999999f);
        hipDeviceSynchronize();
        std::vector<long long> h(waves);
        hipMemcpy(h.data(), dc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        // cycles of SIMD time per instruction = a wave's cycles per iteration / N / waves per SIMD
        printf("  %dw/SIMD %6.2f cyc/instr/wave = %5.2f SIMD-cyc/instr", per_cu, (double) h[waves / 2] / ITER / N,
               (double) h[waves / 2] / ITER / N / per_cu);
        hipFree(dc);
        hipFree(ds);
    }
    printf("\n");
}

int main() {
    run<FMA32>("v_fma_f32");
    run<FMAC32>("v_fmac_f32_e32");
    run<PKFMA32>("v_pk_fma_f32");
    run<PKMUL32>("v_pk_mul_f32");
    run<PKADD32>("v_pk_add_f32");
    run<PKFMA_AND_FMA>("v_pk_fma_f32 / v_fma_f32 alternating");
    run<FMAC_DPP_WAVE>("v_fmac_f32_dpp wave_shr:1");
    run<FMAC_DPP_ROW>("v_fmac_f32_dpp row_shr:1");
    run<MUL_DPP_WAVE>("v_mul_f32_dpp wave_shl:1");
    run<MOV_DPP_WAVE>("v_mov_b32_dpp wave_shr:1");
    run<LSHL>("v_lshlrev_b32 16");
    run<AND>("v_and_b32 0xffff0000");
    run<CVTPK>("v_cvt_pk_bf16_f32");
    run<BFI>("v_bfi_b32");
    run<FMA64>("v_fma_f64");
    return 0;
}
