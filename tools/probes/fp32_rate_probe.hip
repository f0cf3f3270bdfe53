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

enum Op { FMA32, FMAC32, PKFMA32, PKMUL32, FMAC_DPP_WAVE, FMAC_DPP_ROW, MUL_DPP_WAVE, MOV_DPP_WAVE, LSHL, AND, CVTPK, BFI, FMA64, PKFMA_AND_FMA, PKADD32, CVTF32, CVTF32_SDWA, PERM, FMA32_SGPR, MUL32, LSHL_OR, MOV, FMAC_AND_DPP, FMAC_AND_PK, A_FMAC_VV, B_FMAC_SV, C_MUL_V, D_MUL_S, E_FMA_SV, F_LSHL_INPLACE, G_AND_LIT, H_AND_V, I_FMAC_VDEP, J_FMAC_SDEP, K_LSHL_V, L_FMA_VVV_DEP, M_CVTPK_DEP, N_XPASS, O_ZPASS, P_FMA64_S, Q_FMA64_VDEP, R_ADD64, S_MUL64_S, T_FMA64_SDEP, U_ADD64_DEP };

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
            if constexpr (OP == CVTF32) asm volatile("v_cvt_f32_bf16_e32 %0, %1" : "=v"(f[k]) : "v"(u[k1]));
            if constexpr (OP == CVTF32_SDWA)
                asm volatile("v_cvt_f32_bf16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(f[k]) : "v"(u[k1]));
            if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[k]) : "v"(u[k1]), "v"(u[(v + 2) & 7]), "v"(u[(v + 3) & 7]));
            if constexpr (OP == FMA32_SGPR) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "s"(a), "v"(f[k1]));
            if constexpr (OP == MUL32) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[k]) : "s"(a), "v"(f[k1]));
            if constexpr (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(u[k]) : "v"(u[k1]), "v"(u[(v + 2) & 7]));
            if constexpr (OP == MOV) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(u[k]) : "v"(u[k1]));
            if constexpr (OP == FMAC_AND_DPP) {  // the x-pass mix: two DPP-fused of six multiply-adds
                if (v % 3 == 0)
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(f[k]) : "v"(f[k1]), "v"(b));
                else
                    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "s"(a), "v"(f[k1]));
            }
            if constexpr (OP == FMAC_AND_PK) {  // one packed per two plain
                if (v % 3 == 0)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(a2), "v"(b2));
                else
                    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "s"(a), "v"(f[k1]));
            }
            // ---- the same accumulate-in-place dependence (8 chains) for every operand kind ----
            if constexpr (OP == A_FMAC_VV) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "v"(a), "v"(b));
            if constexpr (OP == B_FMAC_SV) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "s"(a), "v"(b));
            if constexpr (OP == C_MUL_V) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(f[k]) : "v"(a));
            if constexpr (OP == D_MUL_S) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(f[k]) : "s"(a));
            if constexpr (OP == E_FMA_SV) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "s"(a), "v"(b));
            if constexpr (OP == F_LSHL_INPLACE) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(u[k]));
            if constexpr (OP == G_AND_LIT) asm volatile("v_and_b32_e32 %0, 0xffff0000, %0" : "+v"(u[k]));
            if constexpr (OP == H_AND_V) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(u[k]) : "v"(u[k1]));
            // ---- second operand produced seven instructions earlier (a neighbouring chain) ----
            if constexpr (OP == I_FMAC_VDEP) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "v"(a), "v"(f[k1]));
            if constexpr (OP == J_FMAC_SDEP) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "s"(a), "v"(f[k1]));
            if constexpr (OP == K_LSHL_V) asm volatile("v_lshlrev_b32_e32 %0, %1, %0" : "+v"(u[k]) : "v"(u[k1]));
            if constexpr (OP == L_FMA_VVV_DEP) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[k]) : "v"(a), "v"(f[k1]), "v"(f[(v + 2) & 7]));
            if constexpr (OP == M_CVTPK_DEP) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[k]) : "v"(f[k]), "v"(f[k1]));
            if constexpr (OP == N_XPASS) {  // the x-pass of a row as kernels_3d_bf16_lanes.hip issues it (8 instructions, two points)
                if (v < 2) {
                    float x0, x1, t0, t1;
                    asm volatile("v_lshlrev_b32_e32 %[x0], 16, %[in]\n\t"
                                 "v_and_b32_e32 %[x1], 0xffff0000, %[in]\n\t"
                                 "v_mul_f32_e32 %[t1], %[c0], %[x0]\n\t"
                                 "v_fmac_f32_e32 %[t1], %[c1], %[x1]\n\t"
                                 "v_mul_f32_dpp %[t0], %[x1], %[c0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "v_fmac_f32_e32 %[t0], %[c1], %[x0]\n\t"
                                 "v_fmac_f32_e32 %[t0], %[c2], %[x1]\n\t"
                                 "v_fmac_f32_dpp %[t1], %[x0], %[c2] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                 : [x0] "=&v"(x0), [x1] "=&v"(x1), [t0] "=&v"(t0), [t1] "=&v"(t1)
                                 : [in] "v"(u[k]), [c0] "v"(a), [c1] "s"(b), [c2] "v"(b));
                    f[k] = t0;
                    f[k1] = t1;
                }
            }
            if constexpr (OP == O_ZPASS) {  // the z-pass of a row (7 instructions, two points), scalar taps
                if (v < 2) {
                    float o0, o1;
                    asm volatile("v_fma_f32 %[o0], %[a2], %[u0], %[cur0]\n\t"
                                 "v_fma_f32 %[o1], %[a2], %[u1], %[cur1]\n\t"
                                 "v_fmac_f32_e32 %[nxt0], %[a1], %[u0]\n\t"
                                 "v_fmac_f32_e32 %[nxt1], %[a1], %[u1]\n\t"
                                 "v_mul_f32_e32 %[cur0], %[a0], %[u0]\n\t"
                                 "v_mul_f32_e32 %[cur1], %[a0], %[u1]\n\t"
                                 "v_cvt_pk_bf16_f32 %[v], %[o0], %[o1]"
                                 : [o0] "=&v"(o0), [o1] "=&v"(o1), [v] "=v"(u[k]), [cur0] "+v"(f[k]), [nxt0] "+v"(f[k1]), [cur1] "+v"(f[(v + 2) & 7]), [nxt1] "+v"(f[(v + 3) & 7])
                                 : [u0] "v"(f[(v + 4) & 7]), [u1] "v"(f[(v + 5) & 7]), [a0] "s"(a), [a1] "s"(b), [a2] "s"(a));
                }
            }
            if constexpr (OP == P_FMA64_S) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "s"(da), "v"(db));
            if constexpr (OP == Q_FMA64_VDEP) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(da), "v"(d[k1]));
            if constexpr (OP == T_FMA64_SDEP) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "s"(da), "v"(d[k1]));
            if constexpr (OP == R_ADD64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[k]) : "v"(db));
            if constexpr (OP == U_ADD64_DEP) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[k]) : "v"(d[k1]), "v"(d[(v + 2) & 7]));
            if constexpr (OP == S_MUL64_S) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[k]) : "s"(da));
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

// what an s_memtime tick is: ticks per 10 ns of s_memrealtime (100 MHz) around a busy loop, beside the shader clock
__global__ void tick_probe(long long *out) {
    float f = threadIdx.x;
    const long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 400000; ++i) asm volatile("v_fmac_f32_e32 %0, %0, %0" : "+v"(f));
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (f == 12345.0f) out[2] = 1;
}

int main() {
    {
        long long *d, h[2];
        hipMalloc(&d, 32);
        hipLaunchKernelGGL(tick_probe, dim3(1024), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(tick_probe, dim3(1024), dim3(256), 0, 0, d);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        int khz = 0;
        hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
        printf("s_memtime: %lld ticks in %lld x 10 ns = %.3f GHz (400000 dependent v_fmac_f32: %.2f ticks each); device clock attribute %d kHz\n",
               h[0], h[1], (double) h[0] / (double) h[1] / 10.0, (double) h[0] / 400000.0, khz);
    }
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
    run<CVTF32>("v_cvt_f32_bf16");
    run<CVTF32_SDWA>("v_cvt_f32_bf16_sdwa WORD_1");
    run<PERM>("v_perm_b32");
    run<FMA32_SGPR>("v_fma_f32 sgpr, v, v (dependent)");
    run<MUL32>("v_mul_f32 sgpr");
    run<LSHL_OR>("v_lshl_or_b32");
    run<MOV>("v_mov_b32");
    run<FMAC_AND_DPP>("1 fmac_dpp : 2 fmac");
    run<FMAC_AND_PK>("1 pk_fma : 2 fmac");
    printf("-- accumulate in place, operand kinds --\n");
    run<A_FMAC_VV>("v_fmac_f32 acc += vA * vB");
    run<B_FMAC_SV>("v_fmac_f32 acc += sA * vB");
    run<C_MUL_V>("v_mul_f32 acc = vA * acc");
    run<D_MUL_S>("v_mul_f32 acc = sA * acc");
    run<E_FMA_SV>("v_fma_f32 acc = sA * vB + acc");
    run<F_LSHL_INPLACE>("v_lshlrev_b32 u <<= 1");
    run<G_AND_LIT>("v_and_b32 u &= literal");
    run<H_AND_V>("v_and_b32 u &= v(other chain)");
    printf("-- an operand produced seven instructions earlier --\n");
    run<I_FMAC_VDEP>("v_fmac_f32 acc += vA * f[k+1]");
    run<J_FMAC_SDEP>("v_fmac_f32 acc += sA * f[k+1]");
    run<K_LSHL_V>("v_lshlrev_b32 u <<= u[k+1]");
    run<L_FMA_VVV_DEP>("v_fma_f32 f = vA * f[k+1] + f[k+2]");
    run<M_CVTPK_DEP>("v_cvt_pk_bf16_f32 u = (f[k], f[k+1])");
    printf("-- fp64 --\n");
    run<FMA64>("v_fma_f64 acc += vA * vB");
    run<P_FMA64_S>("v_fma_f64 acc += sA * vB");
    run<Q_FMA64_VDEP>("v_fma_f64 acc += vA * d[k+1]");
    run<T_FMA64_SDEP>("v_fma_f64 acc += sA * d[k+1]");
    run<R_ADD64>("v_add_f64 acc += vB");
    run<U_ADD64_DEP>("v_add_f64 d = d[k+1] + d[k+2]");
    run<S_MUL64_S>("v_mul_f64 acc *= sA");
    printf("-- kernel sequences (cycles per 16-slot iteration / 16: x-pass = 2 rows of 8, z-pass = 2 rows of 7) --\n");
    run<N_XPASS>("x-pass rows (16 instructions)");
    run<O_ZPASS>("z-pass rows (14 instructions)");
    return 0;
}
