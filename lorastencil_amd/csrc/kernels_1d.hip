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

// ---------------------------------------------------------------------------------------------------------------
// K applications per launch (temporal fusion).  At the reference's size the single sweep is bound
// by launch latency and L2 round trips (~4 us per step whatever the kernel does), so the lever is fewer launches:
// a workgroup loads the window of its 1024 outputs plus 4 K points on either side into LDS once, applies the 9 taps
// K times ping-ponging between two LDS arrays -- the valid range shrinks by 4 points per side and level -- and stores
// level K.  Same taps, same order as the single sweep at every level: bit-identical to K launches.
//
// Boundary semantics of the step-by-step driver: halo cells are never written, so at ODD time levels
// (buffer 1) they hold 0 and at EVEN levels (buffer 0) the caller's input halo.  A fused launch starts at an even
// level; cells outside the interior are therefore forced to 0 at odd intermediate levels and to the source buffer's
// halo value at even ones (the Dirichlet option: the source's halo value at every level).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFusedOut = 1024;  // outputs per workgroup (2^20: 1024 -> 853 GStencils/s, 2048 -> 720, 512 -> 865)

struct ArgsFused1D {
    const double *in;
    double *out;
    int n;           // interior points
    int begin, end;  // interior range of this launch (even begin)
    int dirichlet;
};

template <int K>
__global__ __launch_bounds__(256) void stencil1d_fusedk_kernel(const ArgsFused1D a, const Taps9 W) {
    constexpr int WIN = kFusedOut + 8 * K;
    __shared__ __attribute__((aligned(16))) double L[2][WIN];
    const int tid = threadIdx.x;
    const int t0 = a.begin + (int) blockIdx.x * kFusedOut;  // first output (interior index, even)
    const int w0 = t0 + 4 - 4 * K;                          // padded index of window element 0 (even)
    const int npad = a.n + 8;

    // level 0: the window, 0 where the padded array ends
    for (int i = 2 * tid; i < WIN; i += 512) {
        const int g = w0 + i;
        d2 v;
        if (g >= 0 && g + 1 < npad) {
            v = *reinterpret_cast<const d2 *>(a.in + g);
        } else {
            v.x = (g >= 0 && g < npad) ? a.in[g] : 0.0;
            v.y = (g + 1 >= 0 && g + 1 < npad) ? a.in[g + 1] : 0.0;
        }
        *reinterpret_cast<d2 *>(&L[0][i]) = v;
    }
    __syncthreads();

#pragma unroll
    for (int l = 1; l <= K; ++l) {
        const double *src = L[(l - 1) & 1];
        double *dst = L[l & 1];
        // level l is defined on window elements [4 l, WIN - 4 l)
        constexpr int NPAIR_IT = (WIN - 8 + 511) / 512;  // compile-time trip count: the LDS reads of all iterations overlap
#pragma unroll
        for (int it = 0; it < NPAIR_IT; ++it) {
            const int i = 4 * l + 2 * tid + 512 * it;
            if (i >= WIN - 4 * l) break;
            double win[10];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const d2 v = *reinterpret_cast<const d2 *>(src + i - 4 + 2 * q);
                win[2 * q] = v.x;
                win[2 * q + 1] = v.y;
            }
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                a0 = fma(W.w[t], win[t], a0);
                a1 = fma(W.w[t], win[t + 1], a1);
            }
            const int g = w0 + i;  // padded index of the pair's first point
            if (l < K) {
                // intermediate level: halo cells keep their driver state instead of a computed value
                const bool keep = a.dirichlet || (l & 1) == 0;
                if (!(g >= 4 && g < a.n + 4)) a0 = (keep && g >= 0 && g < npad) ? a.in[g] : 0.0;
                if (!(g + 1 >= 4 && g + 1 < a.n + 4)) a1 = (keep && g + 1 >= 0 && g + 1 < npad) ? a.in[g + 1] : 0.0;
                d2 v;
                v.x = a0;
                v.y = a1;
                *reinterpret_cast<d2 *>(dst + i) = v;
            } else {
                const int o = g - 4;  // interior index
                if (o + 1 < a.end && o + 1 < a.n) {
                    d2 v;
                    v.x = a0;
                    v.y = a1;
                    *reinterpret_cast<d2 *>(a.out + g) = v;
                } else if (o < a.end && o < a.n) {
                    a.out[g] = a0;
                }
            }
        }
        if (l < K) __syncthreads();
    }
}

}  // namespace

hipError_t launch_1d_fused(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    if (begin & 1) return hipErrorInvalidValue;
    ArgsFused1D a;
    a.in = in;
    a.out = out;
    a.n = p.dims[0];
    a.begin = begin;
    a.end = end;
    a.dirichlet = p.boundary == LORA_BC_DIRICHLET;
    Taps9 w;
    for (int t = 0; t < 9; ++t) w.w[t] = p.w[t];
    const long blocks = ((long) end - begin + kFusedOut - 1) / kFusedOut;
    if (p.steps_per_launch == 32)
        hipLaunchKernelGGL((stencil1d_fusedk_kernel<32>), dim3((unsigned) blocks), dim3(256), 0, s, a, w);
    else if (p.steps_per_launch == 16)
        hipLaunchKernelGGL((stencil1d_fusedk_kernel<16>), dim3((unsigned) blocks), dim3(256), 0, s, a, w);
    else if (p.steps_per_launch == 8)
        hipLaunchKernelGGL((stencil1d_fusedk_kernel<8>), dim3((unsigned) blocks), dim3(256), 0, s, a, w);
    else if (p.steps_per_launch == 4)
        hipLaunchKernelGGL((stencil1d_fusedk_kernel<4>), dim3((unsigned) blocks), dim3(256), 0, s, a, w);
    else if (p.steps_per_launch == 2)
        hipLaunchKernelGGL((stencil1d_fusedk_kernel<2>), dim3((unsigned) blocks), dim3(256), 0, s, a, w);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

const char *kernel_name_1d_fused(const Plan &) { return "stencil1d_fusedk_kernel"; }

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
