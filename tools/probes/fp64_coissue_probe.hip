// Probe: do v_mfma_f64_16x16x4_f64 and v_fma_f64 execute CONCURRENTLY on a gfx950 SIMD, or do they share one fp64
// datapath?  (VERDICT r02 "Next round" 1a: one of the four levels of stencil2d_stream_kernel on the matrix pipe beside
// three on the vector pipe only pays if the two pipes overlap for fp64.)
//
//   hipcc --offload-arch=gfx950 -O2 -o fp64_coissue_probe fp64_coissue_probe.hip && ./fp64_coissue_probe
//
// Every wave runs ITER iterations of { NM independent MFMAs, NV independent v_fma_f64 } (inline asm: the order and the
// count are exactly what is written) and stamps s_memtime around the loop.  Printed: shader cycles per iteration (median
// over waves) for one / two / three waves per SIMD, and the wall-clock rate.  If the pipes overlap, {1 MFMA, 16 FMA}
// costs about max(64, 64) cycles; if they share the datapath, about 128.
// Second part: SPLIT roles -- in a 512-thread workgroup waves 0-3 issue only MFMAs and waves 4-7 only FMAs (one of each
// per SIMD): each side's cycles per instruction beside the other vs alone.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int ITER = 500;  // x 4 sub-iterations

template <int NM, int NV>
__device__ __forceinline__ void body(d4 (&acc)[4], double (&x)[16], double a, double b) {
    // interleave: the MFMAs first (they only occupy issue for a few cycles each), then the FMAs
    // four sub-iterations, each on another accumulator (no dependent MFMA chain)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int m = 0; m < NM; ++m) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[(j + 2 * m) & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[v & 15]) : "v"(a), "v"(b));
    }
}

// MODE 0: every wave runs {NM, NV}.  MODE 1: waves 0..3 of a 512-thread workgroup run {NM, 0}, waves 4..7 run {0, NV}.
template <int NM, int NV, int MODE>
__global__ __launch_bounds__(512) void probe(long long *cycles, double *sink, double a, double b) {
    d4 acc[4];
    double x[16];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-9 + i;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int it = 0; it < ITER; ++it) body<NM, NV>(acc, x, a, b);
    } else if (wave < 4) {
        for (int it = 0; it < ITER; ++it) body<NM, 0>(acc, x, a, b);
    } else {
        for (int it = 0; it < ITER; ++it) body<0, NV>(acc, x, a, b);
    }
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int NM, int NV, int MODE>
void run(const char *label, int threads, int blocks_per_cu) {
    int cus = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    const int blocks = cus * blocks_per_cu;
    const int waves = blocks * (threads / 64);
    long long *dc;
    double *ds;
    CHECK(hipMalloc(&dc, sizeof(long long) * waves));
    CHECK(hipMalloc(&ds, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<NM, NV, MODE>), dim3(blocks), dim3(threads), 0, 0, dc, ds, 1.0000001, 0.9999999);
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<NM, NV, MODE>), dim3(blocks), dim3(threads), 0, 0, dc, ds, 1.0000001, 0.9999999);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(waves);
    CHECK(hipMemcpy(h.data(), dc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
    const int wpb = threads / 64;
    auto median = [&](int lo, int hi) {  // waves lo..hi-1 of every block
        std::vector<long long> v;
        for (int b = 0; b < blocks; ++b)
            for (int w = lo; w < hi; ++w) v.push_back(h[b * wpb + w]);
        std::sort(v.begin(), v.end());
        return (double) v[v.size() / 2] / (4 * ITER);
    };
    const double us = ms * 1e3 / reps;
    // s_memtime ticks at a constant 100 MHz on gfx950?  print both the tick count and the wall time per iteration
    if (MODE == 0)
        printf("%-34s waves/SIMD %d  NM %d NV %2d  memtime/iter %8.2f  wall/iter %7.2f ns  (launch %8.1f us)\n", label,
               threads / 256 * blocks_per_cu, NM, NV, median(0, wpb), us * 1e3 / (4 * ITER), us);
    else
        printf("%-34s waves/SIMD %d  NM %d NV %2d  memtime/iter mfma-waves %8.2f  fma-waves %8.2f  wall/iter %7.2f ns\n", label,
               threads / 256 * blocks_per_cu, NM, NV, median(0, 4), median(4, 8), us * 1e3 / (4 * ITER));
    CHECK(hipFree(dc));
    CHECK(hipFree(ds));
}

int main() {
    printf("# ITER %d; one workgroup per CU unless noted; memtime = s_memtime ticks per iteration (median over waves)\n", ITER);
    // one wave per SIMD (256 threads, 1 block per CU)
    run<1, 0, 0>("mfma only", 256, 1);
    run<0, 16, 0>("fma only", 256, 1);
    run<1, 16, 0>("same wave: 1 mfma + 16 fma", 256, 1);
    run<1, 8, 0>("same wave: 1 mfma + 8 fma", 256, 1);
    run<1, 32, 0>("same wave: 1 mfma + 32 fma", 256, 1);
    run<2, 16, 0>("same wave: 2 mfma + 16 fma", 256, 1);
    // two waves per SIMD, same program
    run<1, 0, 0>("mfma only", 512, 1);
    run<0, 16, 0>("fma only", 512, 1);
    run<1, 16, 0>("same wave: 1 mfma + 16 fma", 512, 1);
    run<1, 8, 0>("same wave: 1 mfma + 8 fma", 512, 1);
    // three waves per SIMD
    run<0, 16, 0>("fma only", 256, 3);
    run<1, 0, 0>("mfma only", 256, 3);
    run<1, 16, 0>("same wave: 1 mfma + 16 fma", 256, 3);
    run<1, 5, 0>("same wave: 1 mfma + 5 fma", 256, 3);
    // split roles: one MFMA wave and one FMA wave per SIMD
    run<1, 16, 1>("split waves: mfma | 16 fma", 512, 1);
    run<1, 0, 1>("split waves: mfma | idle", 512, 1);
    run<0, 16, 1>("split waves: idle | 16 fma", 512, 1);
    return 0;
}
