// Probe: what ONE LEVEL of a step of kernels_3d_bf16_lanes.hip costs as a bare instruction sequence on registers (no LDS, no
// memory, no barriers): the x-pass of four rows (32 instructions), the y-pass of two rows, their z-pass and rounding -- in the
// plain form the kernel issues and with the y- and z-passes packed (v_pk_mul_f32 / v_pk_fma_f32 on the lane's column pair).
// One to four waves per SIMD; 256- and 1024-thread workgroups.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o bin/bf16_level_probe bf16_level_probe.hip && ./bin/bf16_level_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 3000;

#define LORA_X1(N) "v_lshlrev_b32_e32 %[x0" #N "], 16, %[in" #N "]\n\t"
#define LORA_X2(N) "v_and_b32_e32 %[x1" #N "], 0xffff0000, %[in" #N "]\n\t"
#define LORA_X3(N) "v_mul_f32_e32 %[t1" #N "], %[c0], %[x0" #N "]\n\t"
#define LORA_X4(N) "v_fmac_f32_e32 %[t1" #N "], %[c1], %[x1" #N "]\n\t"
#define LORA_X5(N) "v_mul_f32_dpp %[t0" #N "], %[x1" #N "], %[c0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LORA_X6(N) "v_fmac_f32_e32 %[t0" #N "], %[c1], %[x0" #N "]\n\t"
#define LORA_X7(N) "v_fmac_f32_e32 %[t0" #N "], %[c2], %[x1" #N "]\n\t"
#define LORA_X8(N) "v_fmac_f32_dpp %[t1" #N "], %[x0" #N "], %[c2] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LORA_XALL(X) X(a) X(b) X(c) X(d)
#define LORA_XROWS LORA_XALL(LORA_X1) LORA_XALL(LORA_X2) LORA_XALL(LORA_X3) LORA_XALL(LORA_X4) LORA_XALL(LORA_X5) LORA_XALL(LORA_X6) LORA_XALL(LORA_X7) LORA_XALL(LORA_X8)

__device__ __forceinline__ void xpass_rows(const unsigned (&in)[4], float c0v, float c1, float c2v, f2 (&t)[4]) {
    float x0a, x1a, x0b, x1b, x0c, x1c, x0d, x1d;
    asm volatile(LORA_XROWS
        : [x0a] "=&v"(x0a), [x1a] "=&v"(x1a), [t0a] "=&v"(t[0].x), [t1a] "=&v"(t[0].y),
          [x0b] "=&v"(x0b), [x1b] "=&v"(x1b), [t0b] "=&v"(t[1].x), [t1b] "=&v"(t[1].y),
          [x0c] "=&v"(x0c), [x1c] "=&v"(x1c), [t0c] "=&v"(t[2].x), [t1c] "=&v"(t[2].y),
          [x0d] "=&v"(x0d), [x1d] "=&v"(x1d), [t0d] "=&v"(t[3].x), [t1d] "=&v"(t[3].y)
        : [ina] "v"(in[0]), [inb] "v"(in[1]), [inc] "v"(in[2]), [ind] "v"(in[3]), [c0] "v"(c0v), [c1] "v"(c1), [c2] "v"(c2v));
}

// MODE 0: plain y / z;  1: packed y / z;  2: x-pass only;  3: plain y / z only;  4: packed y / z only
template <int MODE, int THREADS, int UNROLL = 1>
__global__ __launch_bounds__(THREADS) void probe(long long *cycles, float *sink, float c0, float c1, float c2) {
    unsigned in[4];
    f2 cur[2], nxt[2];
    for (int r = 0; r < 4; ++r) in[r] = 0x3f803f80u + threadIdx.x * 7 + r;
    for (int r = 0; r < 2; ++r) cur[r] = nxt[r] = (f2){0.5f, 0.25f};
    float b0 = c0, b1 = c1, b2 = c2, a0 = c0, a1 = c1, a2 = c2;
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(a0), "+v"(a1), "+v"(a2));
    const f2 b0p = {b0, b0}, b1p = {b1, b1}, b2p = {b2, b2}, a0p = {a0, a0}, a1p = {a1, a1}, a2p = {a2, a2};
    f2 t[4];
    for (int r = 0; r < 4; ++r) t[r] = (f2){0.1f * r, 0.2f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER / UNROLL; ++it) {
#pragma unroll
      for (int un = 0; un < UNROLL; ++un) {
        if constexpr (MODE <= 2) xpass_rows(in, c0, c1, c2, t);
        if constexpr (MODE != 2) {
#pragma unroll
            for (int r = 1; r < 3; ++r) {
                unsigned v;
                if constexpr (MODE == 0 || MODE == 3) {
                    float u0 = b0 * t[r - 1].x, u1 = b0 * t[r - 1].y;
                    asm volatile("v_fmac_f32_e32 %0, %2, %3\n\tv_fmac_f32_e32 %1, %2, %4\n\tv_fmac_f32_e32 %0, %5, %6\n\tv_fmac_f32_e32 %1, %5, %7"
                                 : "+v"(u0), "+v"(u1) : "v"(b1), "v"(t[r].x), "v"(t[r].y), "v"(b2), "v"(t[r + 1].x), "v"(t[r + 1].y));
                    float o0, o1;
                    asm volatile("v_fma_f32 %[o0], %[a2], %[u0], %[cur0]\n\t"
                                 "v_fma_f32 %[o1], %[a2], %[u1], %[cur1]\n\t"
                                 "v_fmac_f32_e32 %[nxt0], %[a1], %[u0]\n\t"
                                 "v_fmac_f32_e32 %[nxt1], %[a1], %[u1]\n\t"
                                 "v_mul_f32_e32 %[cur0], %[a0], %[u0]\n\t"
                                 "v_mul_f32_e32 %[cur1], %[a0], %[u1]\n\t"
                                 "v_cvt_pk_bf16_f32 %[v], %[o0], %[o1]"
                                 : [o0] "=&v"(o0), [o1] "=&v"(o1), [v] "=v"(v), [cur0] "+v"(cur[r - 1].x), [nxt0] "+v"(nxt[r - 1].x),
                                   [cur1] "+v"(cur[r - 1].y), [nxt1] "+v"(nxt[r - 1].y)
                                 : [u0] "v"(u0), [u1] "v"(u1), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2));
                } else {
                    f2 u, o;
                    asm volatile("v_pk_mul_f32 %[u], %[b0], %[tm]\n\t"
                                 "v_pk_fma_f32 %[u], %[b1], %[t0], %[u]\n\t"
                                 "v_pk_fma_f32 %[u], %[b2], %[tp], %[u]\n\t"
                                 "v_pk_fma_f32 %[o], %[a2], %[u], %[cur]\n\t"
                                 "v_pk_fma_f32 %[nxt], %[a1], %[u], %[nxt]\n\t"
                                 "v_pk_mul_f32 %[cur], %[a0], %[u]"
                                 : [u] "=&v"(u), [o] "=&v"(o), [cur] "+v"(cur[r - 1]), [nxt] "+v"(nxt[r - 1])
                                 : [b0] "v"(b0p), [b1] "v"(b1p), [b2] "v"(b2p), [tm] "v"(t[r - 1]), [t0] "v"(t[r]), [tp] "v"(t[r + 1]), [a0] "v"(a0p),
                                   [a1] "v"(a1p), [a2] "v"(a2p));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v) : "v"(o.x), "v"(o.y));
                }
                in[r] = v;
            }
        }
      }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 4; ++r) s += t[r].x + t[r].y + (float) in[r];
    for (int r = 0; r < 2; ++r) s += cur[r].x + cur[r].y + nxt[r].x + nxt[r].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int THREADS, int UNROLL = 1>
void run(const char *label, int instrs) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%-44s %4d-thread WGs, %2d instructions:", label, THREADS, instrs);
    for (int wps : {1, 2, 4}) {
        const int per_cu = wps * 256 / THREADS;
        if (per_cu < 1) continue;
        const int blocks = prop.multiProcessorCount * per_cu, waves = blocks * (THREADS / 64);
        long long *dc;
        float *ds;
        hipMalloc(&dc, sizeof(long long) * waves);
        hipMalloc(&ds, 8);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, THREADS, UNROLL>), dim3(blocks), dim3(THREADS), 0, 0, dc, ds, 0.25f, 0.5f, 0.25f);
        hipDeviceSynchronize();
        std::vector<long long> h(waves);
        hipMemcpy(h.data(), dc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double) h[waves / 2] / (ITER / UNROLL * UNROLL);
        printf("  %dw/SIMD %7.1f cyc/iter/wave = %5.2f SIMD-cyc/instr", wps, cyc, cyc / instrs / wps);
        hipFree(dc);
        hipFree(ds);
    }
    printf("\n");
}

int main() {
    run<0, 256>("x-pass 4 rows + plain y / z of 2 rows", 32 + 2 * 13);
    run<1, 256>("x-pass 4 rows + packed y / z of 2 rows", 32 + 2 * 7);
    run<2, 256>("x-pass 4 rows", 32);
    run<3, 256>("plain y / z of 2 rows", 2 * 13);
    run<4, 256>("packed y / z of 2 rows", 2 * 7);
    run<0, 1024>("x-pass 4 rows + plain y / z of 2 rows", 32 + 2 * 13);
    run<1, 1024>("x-pass 4 rows + packed y / z of 2 rows", 32 + 2 * 7);
    printf("-- the same body unrolled: does a longer instruction stream cost more per instruction? --\n");
    run<0, 1024, 2>("plain, loop body x 2 (~0.7 KB)", 58);
    run<0, 1024, 8>("plain, loop body x 8 (~2.8 KB)", 58);
    run<0, 1024, 24>("plain, loop body x 24 (~8 KB)", 58);
    run<0, 1024, 60>("plain, loop body x 60 (~20 KB)", 58);
    run<0, 1024, 150>("plain, loop body x 150 (~50 KB)", 58);
    return 0;
}
