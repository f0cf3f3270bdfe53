// kernels_1d.hip -- 1D sweep for gfx950: out[i+4] = sum_{t<9} w[t] * in[i+t], i in [begin, end), on a padded
// array of n+8 doubles.  Replaces kernel_1d1r / kernel_1d2r (1d/gpu_1r.cu:21-87, 1d/gpu_2r.cu:22-88), which are
// the same 9-tap algorithm with different weights.
//
// The reference views 1024 points as an 8 x 128 matrix to feed 8x8x4 tensor-core tiles; that shape has no
// counterpart here.  At the reference's size (N = 2^20, 8 MB) the sweep lives in L2 / Infinity Cache and is
// launch-latency bound, so the kernel is the plain bandwidth form: each lane owns 2 adjacent points, loads its
// 10-wide window with five aligned 16-byte loads (neighbouring lanes overlap in L1), applies the 9 taps in tap
// order (the order of the reference's CPU check, 1d/main.cu:34-40) with fused multiply-adds and writes 16 bytes.
#include <hip/hip_runtime.h>

#include "device_common.h"

namespace lora {

namespace {

__global__ __launch_bounds__(256) void stencil1d_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                        int begin, int end, const Taps9 W) {
    const long pair = (long) blockIdx.x * 256 + threadIdx.x;
    const long i = begin + 2 * pair;  // begin is even (checked on the host)
    if (i >= end) return;
    if (i + 1 < end) {
        double win[10];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const d2 v = *reinterpret_cast<const d2 *>(in + i + 2 * q);
            win[2 * q] = v.x;
            win[2 * q + 1] = v.y;
        }
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            a0 = fma(W.w[t], win[t], a0);
            a1 = fma(W.w[t], win[t + 1], a1);
        }
        d2 r;
        r.x = a0;
        r.y = a1;
        *reinterpret_cast<d2 *>(out + i + 4) = r;
    } else {
        // odd tail: one point, scalar loads stay inside the padded array
        double a0 = 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) a0 = fma(W.w[t], in[i + t], a0);
        out[i + 4] = a0;
    }
}

}  // namespace

hipError_t launch_1d(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    if (begin & 1) return hipErrorInvalidValue;
    Taps9 w;
    for (int t = 0; t < 9; ++t) w.w[t] = p.w[t];
    const long pairs = ((long) end - begin + 1) / 2;
    const long blocks = (pairs + 255) / 256;
    hipLaunchKernelGGL(stencil1d_kernel, dim3((unsigned) blocks), dim3(256), 0, s, in, out, begin, end, w);
    return hipGetLastError();
}

const char *kernel_name_1d(const Plan &) { return "stencil1d_kernel"; }

}  // namespace lora
