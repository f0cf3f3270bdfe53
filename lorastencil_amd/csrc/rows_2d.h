// rows_2d.h -- evaluation of one input row of a radius-3 2D stencil into register accumulators (shared by the
// temporally fused 2D kernels: kernels_2d_fused.hip, kernels_2d_stream.hip).
#pragma once

#include "device_common.h"

namespace lora {

// Evaluation of one input row's contribution: EVAL 0..2 = direct taps of the diamond / star / box tap set;
// EVAL_LR_* = the paper's low-rank form on the VECTOR pipe: per factor term t a horizontal pass h_t = v_t * row (once
// per input row and column), then u_t[dy] h_t scattered to the output rows -- (|v_t| + |u_t|) instead of
// |u_t| |v_t| multiply-adds per column.  With the reference's factors that is 34 instead of 50 operations per input
// row for star2d1r (rank 1 on the support 1..5 plus its 8-point correction, 2d/gpu.cu:486-487, :249-264) and 60
// instead of 98 for the box tables (pyramid terms live on the nested supports 0..6, 1..5, 2..4, 2d/gpu.cu:280-350).
// EVAL_LR_PYRAMID_SYM: the pyramid form for mirror-symmetric horizontal profiles (v_t[dx] = v_t[6-dx], true for every
// table the pyramid factoriser accepts): the mirrored taps of a row are added once (3 adds per column) and shared by
// the three terms -- 12 instead of 15 operations per column for the horizontal pass, 54 instead of 60 per input row.
// EVAL_LR_PYRAMID_SYM_GAP: additionally the middle term vanishes next to its centre (u_1[2] = u_1[4] = v_1[2] = 0), as in
// the reference's own table, whose second factor is (0, 1, 0, -1, 0, 1, 0) (2d/main.cu:151-167 through 2d/gpu.cu:280-350):
// those taps are skipped at compile time, 48 operations per input row.
// EVAL_NEST: tables whose rows are multiples of NESTED mirror-symmetric profiles -- row dy is g[k] T_k with
// k = 3 - |dy - 3| and T_0 = x3, T_k = a_k T_(k-1) + (x_(3-k) + x_(3+k)).  The reference's star2d1r table is one
// (2d/main.cu:187-195: rows (1), 2 (1 2 1), 2 (1 2 4 2 1), (1 4 8 16 8 4 1): a = (2, 2, 4), g = (1, 2, 2, 1)), which
// makes a row cost 6 operations for the three profiles plus 7 scattered multiply-adds: 13 per column instead of the 17
// of the rank-1 + correction form (25 direct).  u[0][0..3] = g, v[0][1..3] = a.
enum { EVAL_LR_DIAMOND = 3, EVAL_LR_PYRAMID = 4, EVAL_LR_PYRAMID_SYM = 5, EVAL_LR_PYRAMID_SYM_GAP = 6, EVAL_NEST = 7 };

struct LowRankTaps {
    double u[3][7];  // vertical profiles
    double v[3][7];  // horizontal profiles
    double rc;       // LR_DIAMOND: +rc on the four axis tips, -rc on the four (+-2, +-2) corners
};

// ROT rotates the accumulator index: logical output row r lives in acc[(r + ROT) % R] (rotating register files of the
// row-streaming kernel; 0 = the plain tile kernels).
template <int EVAL, int R, int ROT = 0>
__device__ __forceinline__ void apply_row(const int j, const double (&win)[8], double (&acc0)[R], double (&acc1)[R],
                                          const Taps49 &W, const LowRankTaps &F) {
    if constexpr (EVAL <= TAPS2D_BOX) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int dy = j - r;
            if (dy >= 0 && dy < 7) {
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    if (tap_on<EVAL>(dy, dx)) {
                        const double wt = W.w[dy * 7 + dx];
                        acc0[(r + ROT) % R] = fma(wt, win[dx], acc0[(r + ROT) % R]);
                        acc1[(r + ROT) % R] = fma(wt, win[dx + 1], acc1[(r + ROT) % R]);
                    }
                }
            }
        }
    } else if constexpr (EVAL == EVAL_NEST) {
        double t0[4], t1[4];  // the four profiles of this row for column 0 / column 1
        t0[0] = win[3];
        t1[0] = win[4];
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            t0[k] = fma(F.v[0][k], t0[k - 1], win[3 - k] + win[3 + k]);
            t1[k] = fma(F.v[0][k], t1[k - 1], win[4 - k] + win[4 + k]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int dy = j - r;
            if (dy >= 0 && dy < 7) {
                const int k = dy <= 3 ? dy : 6 - dy;
                acc0[(r + ROT) % R] = fma(F.u[0][k], t0[k], acc0[(r + ROT) % R]);
                acc1[(r + ROT) % R] = fma(F.u[0][k], t1[k], acc1[(r + ROT) % R]);
            }
        }
    } else if constexpr (EVAL == EVAL_LR_PYRAMID_SYM || EVAL == EVAL_LR_PYRAMID_SYM_GAP) {
        constexpr bool GAP = EVAL == EVAL_LR_PYRAMID_SYM_GAP;
        // s[k] = win[k] + win[6 - k] for column 0, win[k + 1] + win[7 - k] for column 1 (k = 0..2); centre taps as they are
        double s0[3], s1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s0[k] = win[k] + win[6 - k];
            s1[k] = win[k + 1] + win[7 - k];
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int lo = t, hi = 6 - lo;  // support of term t (nested: 0..6, 1..5, 2..4)
            double h0 = F.v[t][3] * win[3], h1 = F.v[t][3] * win[4];
#pragma unroll
            for (int k = 2; k >= lo; --k) {
                if (GAP && t == 1 && k == 2) continue;
                h0 = fma(F.v[t][k], s0[k], h0);
                h1 = fma(F.v[t][k], s1[k], h1);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int dy = j - r;
                if (dy >= lo && dy <= hi && !(GAP && t == 1 && (dy == 2 || dy == 4))) {
                    acc0[(r + ROT) % R] = fma(F.u[t][dy], h0, acc0[(r + ROT) % R]);
                    acc1[(r + ROT) % R] = fma(F.u[t][dy], h1, acc1[(r + ROT) % R]);
                }
            }
        }
    } else {
        constexpr int NT = (EVAL == EVAL_LR_DIAMOND) ? 1 : 3;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int lo = (EVAL == EVAL_LR_DIAMOND) ? 1 : t, hi = 6 - lo;  // support of term t, both directions
            double h0 = F.v[t][lo] * win[lo], h1 = F.v[t][lo] * win[lo + 1];
#pragma unroll
            for (int dx = lo + 1; dx <= hi; ++dx) {
                h0 = fma(F.v[t][dx], win[dx], h0);
                h1 = fma(F.v[t][dx], win[dx + 1], h1);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int dy = j - r;
                if (dy >= lo && dy <= hi) {
                    acc0[(r + ROT) % R] = fma(F.u[t][dy], h0, acc0[(r + ROT) % R]);
                    acc1[(r + ROT) % R] = fma(F.u[t][dy], h1, acc1[(r + ROT) % R]);
                }
            }
        }
        if constexpr (EVAL == EVAL_LR_DIAMOND) {
            // the 8-point correction, sharing the sums of mirrored taps between the rows that use them
            const double s06_0 = win[0] + win[6], s06_1 = win[1] + win[7];  // (dy = 3, dx = 0 and 6)
            const double s15_0 = win[1] + win[5], s15_1 = win[2] + win[6];  // (dy = 1 or 5, dx = 1 and 5)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int dy = j - r;
                if (dy == 0 || dy == 6) {
                    acc0[(r + ROT) % R] = fma(F.rc, win[3], acc0[(r + ROT) % R]);
                    acc1[(r + ROT) % R] = fma(F.rc, win[4], acc1[(r + ROT) % R]);
                } else if (dy == 3) {
                    acc0[(r + ROT) % R] = fma(F.rc, s06_0, acc0[(r + ROT) % R]);
                    acc1[(r + ROT) % R] = fma(F.rc, s06_1, acc1[(r + ROT) % R]);
                } else if (dy == 1 || dy == 5) {
                    acc0[(r + ROT) % R] = fma(-F.rc, s15_0, acc0[(r + ROT) % R]);
                    acc1[(r + ROT) % R] = fma(-F.rc, s15_1, acc1[(r + ROT) % R]);
                }
            }
        }
    }
}

}  // namespace lora
