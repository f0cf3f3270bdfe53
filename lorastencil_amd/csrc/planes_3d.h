// planes_3d.h -- the per-plane tap evaluation shared by the fp64 3D kernels that fuse applications in time
// (kernels_3d_fused.hip, kernels_3d_planes.hip): a plane of level l - 1 is read once from LDS and scattered, weighted
// per dz / dy / dx, into three rotating accumulator sets of level l (the planes above, at and below it).  The order in
// which a cell receives its taps -- dz outermost, then dy, then dx -- is the single-sweep kernel's (kernels_3d.hip), so
// any chain of fused levels is bit-identical to the same number of single sweeps.
#pragma once

#include <hip/hip_runtime.h>

#include "device_common.h"

namespace lora {

template <int TAPSET>
__host__ __device__ constexpr bool tap_on3(int dz, int dy, int dx) {
    return TAPSET == TAPS3D_BOX ? true : (((dz != 1) + (dy != 1) + (dx != 1)) <= 1);
}

// first tap (in application order) of the dz = 0 group: the tap that opens a result plane's accumulation
template <int TAPSET>
__host__ __device__ constexpr int first_tap3() {
    for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
            if (tap_on3<TAPSET>(0, dy, dx)) return dy * 3 + dx;
    return 0;
}

// Row `j` of a strip's window (6 doubles, the lane's columns are elements 2 and 3) added, weighted per dz / dy / dx,
// to the rotating accumulator sets: a plane of phase PH feeds result plane slot (PH - dz) mod 3.
// FRESH: the slot of the dz = 0 group starts a new result plane -- its first tap is fma(w, x, 0) with a literal zero
// (the same value as adding to a zeroed register, bit for bit), so the caller never has to clear a published slot.
template <int TAPSET, int RY, int PH, bool FRESH = false>
__device__ __forceinline__ void scatter_row(double (&x0)[3][RY], double (&x1)[3][RY], const double (&win)[6], int j,
                                            const Taps27 &W) {
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
        const int s = (PH - dz + 3) % 3;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int dy = j - r;
            if (dy >= 0 && dy < 3) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (tap_on3<TAPSET>(dz, dy, dx)) {
                        const double wt = W.w[dz * 9 + dy * 3 + dx];
                        if (FRESH && dz == 0 && dy * 3 + dx == first_tap3<TAPSET>()) {
                            x0[s][r] = fma(wt, win[dx + 1], 0.0);
                            x1[s][r] = fma(wt, win[dx + 2], 0.0);
                        } else {
                            x0[s][r] = fma(wt, win[dx + 1], x0[s][r]);
                            x1[s][r] = fma(wt, win[dx + 2], x1[s][r]);
                        }
                    }
                }
            }
        }
    }
}

// Exactly separable taps w[dz][dy][dx] = a[dz] b[dy] c[dx] (the reference's box: taps depend on dx only): a plane is
// filtered along x (3 taps per row of the window), along y (3 rows per cell) and its result added, weighted by a[dz], to
// the three rotating sets -- 9-10 fused multiply-adds per cell instead of 27.  The summation order differs from the
// 27-tap form (same value while every partial sum is an exact integer, ~1 ulp per application otherwise).
// cba = {c[0..2], b[0..2], a[0..2]} in W.w[0..8].
template <int RY, int PH, bool FRESH>
__device__ __forceinline__ void scatter_plane_sep(double (&x0)[3][RY], double (&x1)[3][RY], const double (&win)[RY + 2][6],
                                                  const Taps27 &W) {
    double t0[RY + 2], t1[RY + 2];
#pragma unroll
    for (int j = 0; j < RY + 2; ++j) {
        t0[j] = fma(W.w[2], win[j][3], fma(W.w[1], win[j][2], W.w[0] * win[j][1]));
        t1[j] = fma(W.w[2], win[j][4], fma(W.w[1], win[j][3], W.w[0] * win[j][2]));
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const double u0 = fma(W.w[5], t0[r + 2], fma(W.w[4], t0[r + 1], W.w[3] * t0[r]));
        const double u1 = fma(W.w[5], t1[r + 2], fma(W.w[4], t1[r + 1], W.w[3] * t1[r]));
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            const int s = (PH - dz + 3) % 3;
            if (FRESH && dz == 0) {
                x0[s][r] = W.w[6] * u0;
                x1[s][r] = W.w[6] * u1;
            } else {
                x0[s][r] = fma(W.w[6 + dz], u0, x0[s][r]);
                x1[s][r] = fma(W.w[6 + dz], u1, x1[s][r]);
            }
        }
    }
}

}  // namespace lora
