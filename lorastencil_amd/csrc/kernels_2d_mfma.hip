// kernels_2d_mfma.hip -- the paper's low-rank formulation of the 2D sweeps on CDNA4 matrix cores:
//     Out = sum_t (U_t . X) . V_t  (+ a few residual taps on the vector pipe)
// with U_t / V_t the banded (Toeplitz) matrices of the column / row factors u_t, v_t of the 7x7 weight matrix
// (2d/gpu.cu:353-369, :494-503) and X a window of the LDS-staged tile.  Replaces the tensor-core path of
// kernel2d_{star2d1r,box2d3r,star2d3r} (2d/gpu.cu:31-273); re-tiled for v_mfma_f64_16x16x4_f64 on 64-wide waves.
//
// Register-level reuse of the first product (the reference's key trick, 2d/gpu.cu:244), CDNA4 form:
//   v_mfma_f64_16x16x4: A[i][k] on lane (i + 16k), B[k][j] on lane (j + 16k), D[g + 4r][j] in register r of
//   lane (j + 16g).  The first product is computed TRANSPOSED, T^T = X^T . U^T (A = a 4-row x 16-column window of
//   the tile read straight from LDS, B = band of u), so register r of lane (i + 16g) holds T[i][4r + g]: exactly
//   the A operand of k-step r of the second product Out = T . V.  No LDS round trip, no shuffle, and -- unlike the
//   m8n8k4 layout of the reference -- no permutation of V's rows (2d/gpu.cu:506-519 is not needed).
//
// Geometry: a 256-thread workgroup stages a 40 x 144 window for a 32 x 128 output tile; wave (wr, wc) owns the
// 16 x 64 sub-tile at (16 wr, 64 wc): per factor term 5 T-blocks x 6 k-steps (24 input rows, 22 used) and
// 4 output blocks x 6 k-steps (22 of 24 T columns used) = 54 MFMAs; the LDS row pitch of 144 doubles makes the
// A-fragment reads (16 lanes per row, 4 rows) bank-conflict free.
//
// This variant exists to evidence the MFMA-vs-vector choice: on MI355X the fp64 matrix rate equals the
// fp64 vector rate and the banded operands are 70 % zeros, so it issues 3.4 (rank 1) to 10 (rank 3) MFMA cycles per
// point against 1.6 to 3.1 FMA cycles for the direct kernel.
#include <hip/hip_runtime.h>

#include <cstring>

#include "device_common.h"

namespace lora {

namespace {

constexpr int kTH = 32;                  // output rows per workgroup
constexpr int kTW = 128;                 // output columns per workgroup
constexpr int kLH = kTH + 8;             // staged rows: 3 + 32 + 3, rounded up to the 24-row K extent of wave row 1
constexpr int kLW = kTW + 16;            // staged columns: T block 4 of the right-hand waves reads up to column 143
constexpr int kChunks = kLW / 2;         // 16-byte chunks per staged row
constexpr int kNChunk = kLH * kChunks;   // 2880
constexpr int kNIT = (kNChunk + 255) / 256;

struct ArgsMfma {
    const double *in;
    double *out;
    int ld, m, n;
    int row_begin, row_end;
    int tiles_x, tiles_y, panel_w;
};

// Factor tables as the kernel wants them: zero-padded so that a band entry is a plain indexed read.
//   ub[t][15 + e] = u_t[e], vb[t][16 + e] = v_t[e] for e = 0..6, zero elsewhere (40 entries each).
struct BandsDev {
    double ub[3][40];
    double vb[3][40];
    int nresid;
    int rdy[16], rdx[16];
    double rw[16];
};

template <int RANK>
__global__ __launch_bounds__(256, 3) void stencil2d_mfma_kernel(const ArgsMfma a, const BandsDev bands) {
    __shared__ __attribute__((aligned(16))) double tile[kLH * kLW];
    __shared__ double sb[2 * 3 * 40];

    const int tid = threadIdx.x;
    int ty, tx;
    panel_major(xcd_contiguous(blockIdx.x, gridDim.x), a.tiles_x, a.tiles_y, a.panel_w, ty, tx);
    const int i0 = a.row_begin + ty * kTH;
    const int j0 = tx * kTW;

    // ---- stage the window: padded rows i0+1 .. i0+40, padded columns j0 .. j0+143 (clamped) -------------
    {
        d2 stage[kNIT];
        const int max_row = a.m + 7, max_col = a.n + 6;
#pragma unroll
        for (int it = 0; it < kNIT; ++it) {
            const int k = tid + it * 256;
            if (k < kNChunk) {
                const int r = k / kChunks, c = k - r * kChunks;
                const int gr = min(i0 + 1 + r, max_row), gc = min(j0 + 2 * c, max_col);
                stage[it] = *reinterpret_cast<const d2 *>(a.in + (size_t) gr * a.ld + gc);
            }
        }
        if (tid < 240) sb[tid] = (tid < 120) ? bands.ub[0][tid] : bands.vb[0][tid - 120];
#pragma unroll
        for (int it = 0; it < kNIT; ++it) {
            const int k = tid + it * 256;
            if (k < kNChunk) *reinterpret_cast<d2 *>(tile + 2 * k) = stage[it];
        }
    }
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int li = lane & 15;  // row of A / column of B and D
    const int lg = lane >> 4;  // k index of A and B, row group of D
    // window origin of this wave inside the tile: tile row 16 wr (= output row 0 minus 3), tile column 64 wc
    const double *wtile = tile + (16 * wr) * kLW + 64 * wc;

    d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};

#pragma unroll
    for (int t = 0; t < RANK; ++t) {
        // band fragments of this term: k-step s covers band offsets 4s .. 4s+3
        double ufrag[6], vfrag[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            ufrag[s] = sb[t * 40 + 15 + 4 * s + lg - li];        // u_t[4s + g - i]
            vfrag[s] = sb[120 + t * 40 + 16 + 4 * s + lg - li - 1];  // v_t[4s + g - j - 1]
        }
        // first product, transposed: Tt[q] = X^T(16 cols x 24 rows) . U^T(24 x 16)
        d4 Tt[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            Tt[q] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const double x = wtile[(4 * s + lg) * kLW + 16 * q + li];
                Tt[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, ufrag[s], Tt[q], 0, 0, 0);
            }
        }
        // second product: Out[q] += T[:, 16q .. 16q+23] . V ; register r of Tt is k-step r of its block
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Tt[q][s], vfrag[s], acc[q], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Tt[q + 1][s], vfrag[4 + s], acc[q], 0, 0, 0);
        }
    }

    // ---- residual taps on the vector pipe (star2d1r: the reference's 8-point correction, 2d/gpu.cu:249-264) ----
    // D layout: register r of lane (j + 16 g) is output (row g + 4r, column 16 q + j) of the wave's sub-tile.
    const int nres = bands.nresid;
    for (int k = 0; k < nres; ++k) {
        const int dy = bands.rdy[k], dx = bands.rdx[k];
        const double w = bands.rw[k];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[q][r] = fma(w, wtile[(lg + 4 * r + 3 + dy) * kLW + 16 * q + li + 4 + dx], acc[q][r]);
    }

    // ---- store: each instruction writes four 128-byte row segments -----------------------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = j0 + 64 * wc + 16 * q + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + 16 * wr + lg + 4 * r;
            if (col < a.n && row < a.row_end) a.out[(size_t) (row + 4) * a.ld + (col + 4)] = acc[q][r];
        }
    }
}

// The band tables travel as a kernel argument (2.2 KB of the 4 KB kernarg segment): no device allocation, no upload,
// nothing to synchronise when a plan's factors change, and the launch can be captured into a hipGraph.
BandsDev make_bands(const LowRank2D &lr) {
    BandsDev h{};
    for (int t = 0; t < lr.rank; ++t)
        for (int e = 0; e < 7; ++e) {
            h.ub[t][15 + e] = lr.u[t][e];
            h.vb[t][16 + e] = lr.v[t][e];
        }
    h.nresid = lr.nresid;
    for (int k = 0; k < lr.nresid; ++k) {
        h.rdy[k] = lr.rdy[k];
        h.rdx[k] = lr.rdx[k];
        h.rw[k] = lr.rw[k];
    }
    return h;
}

}  // namespace

hipError_t launch_2d_mfma(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (!p.lowrank_valid) return hipErrorNotSupported;
    const BandsDev bands = make_bands(p.lowrank);
    ArgsMfma a;
    a.in = in;
    a.out = out;
    a.m = p.dims[0];
    a.n = p.dims[1];
    a.ld = a.n + 8;
    a.row_begin = begin;
    a.row_end = end;
    a.tiles_x = (a.n + kTW - 1) / kTW;
    a.tiles_y = (end - begin + kTH - 1) / kTH;
    a.panel_w = p.panel_width < 1 ? 1 : (p.panel_width > a.tiles_x ? a.tiles_x : p.panel_width);
    const long nblocks = (long) a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    const dim3 grid((unsigned) nblocks), block(256);
    switch (p.lowrank.rank) {
        case 1:
            hipLaunchKernelGGL(stencil2d_mfma_kernel<1>, grid, block, 0, s, a, bands);
            break;
        case 2:
            hipLaunchKernelGGL(stencil2d_mfma_kernel<2>, grid, block, 0, s, a, bands);
            break;
        case 3:
            hipLaunchKernelGGL(stencil2d_mfma_kernel<3>, grid, block, 0, s, a, bands);
            break;
        default:
            return hipErrorNotSupported;
    }
    return hipGetLastError();
}

const char *kernel_name_2d_mfma(const Plan &) { return "stencil2d_mfma_kernel"; }

}  // namespace lora
