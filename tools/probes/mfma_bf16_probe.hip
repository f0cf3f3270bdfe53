// Probe of v_mfma_f32_16x16x32_bf16 operand layouts on gfx950: D = A (16x32) x B (32x16), fp32 accumulate.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_bf16_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D) {
    const int l = threadIdx.x, m = l & 15, kb = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16) A[m * 32 + 8 * kb + j];      // assumed: lane (m, kb) holds A[m][8 kb + j]
        b[j] = (__bf16) B[(8 * kb + j) * 16 + m];    // assumed: lane (n, kb) holds B[8 kb + j][n]
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * kb + i) * 16 + m] = c[i];  // assumed: lane (n, g) holds D[4 g + i][n]
}
int main() {
    float hA[512], hB[512], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float) ((i * 7) % 13 - 6); hB[i] = (float) ((i * 5) % 11 - 5); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("mfma_f32_16x16x32_bf16 layout probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return bad != 0;
}
