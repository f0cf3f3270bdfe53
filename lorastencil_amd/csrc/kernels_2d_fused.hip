// kernels_2d_fused.hip -- TWO kernel applications of a radius-3 2D stencil per launch (temporal fusion).
//
// The single-sweep kernel (kernels_2d.hip) moves exactly the compulsory 16 B per point per application through
// HBM (PMC: 1.007 x, profiles/) and runs at ~70 % of the 8 TB/s peak, i.e. at what this chip's HBM delivers to a
// streaming kernel.  The only lever left is fewer bytes per application: here one launch reads the grid once,
// applies the stencil twice with the intermediate time level kept in LDS, and writes once -- 8 B per point per
// application.  The reference has no in-kernel temporal blocking; its x3 GStencil/s factor
// only *accounts* one radius-3 launch as three radius-1 steps.
//
// Semantics are those of two consecutive launches of the reference driver, including its de-facto
// boundary condition: the intermediate level is "buffer 1", whose halo cells are never written and hold 0.  A fused
// launch therefore always starts at an EVEN step: it reads a buffer whose halo is the caller's input halo, treats
// every intermediate cell outside the interior as 0, and writes the interior of the other buffer (the driver in
// capi.cpp keeps the halos of the two physical buffers in the state this needs).
//
// Geometry (256 threads, 4 waves; R1 = intermediate rows per wave, default 8):
//   output tile        TH = 4 R1 - 6 rows x 122 columns          (61 lanes x 2 columns; j0 = 122 tx is even)
//   intermediate tile  4 R1 rows      x 128 columns  in LDS (B)  (64 lanes x 2 columns, starts 3 left of the output)
//   input window       4 R1 + 6 rows  x 136 columns  in LDS (A)  (starts 6 left: an even column, so every global
//                                                                 load is a 16-byte aligned piece of a row)
// Shifting each level's lane->column map by 3 makes every 7-tap window of a lane's column pair an ALIGNED 8-wide
// LDS window: 4 x ds_read_b128 per row and level (the single-sweep kernel needs 5).
// Cost of fusing: the intermediate tile is 1.29 x the output tile (recomputed halo), 30 instead of 25 FMAs per
// point and application.  The intermediate tile overwrites the input window in LDS once it has been consumed
// (41 KiB per workgroup -> 3 workgroups per CU).
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "rows_2d.h"

namespace lora {

namespace {

constexpr int kOutW = 122;            // output columns per tile
constexpr int kMidW = 128;            // intermediate columns per tile
constexpr int kInW = 136;             // staged input columns per tile
constexpr int kInChunks = kInW / 2;   // 16-byte chunks per staged row

struct ArgsFused {
    const double *in;
    double *out;
    int ld, m, n;
    int row_begin, row_end;
    int tiles_x, tiles_y, panel_w;
    int dirichlet;  // intermediate cells outside the interior keep the input halo value instead of 0
    int ablate;     // timing-only diagnostics: 1 = no stores, 2 = no window loads (results are then wrong)
};

// PERSIST: the grid is 3 workgroups per CU; each walks its XCD's run of tiles and fetches the next tile's input
// window into registers while the current tile is being computed, so the HBM latency of a tile is hidden behind
// the two applications of the previous one instead of behind the other resident workgroup only.
template <int TAPSET, int R1, bool PERSIST>
__global__ __launch_bounds__(256, 3) void stencil2d_fused2_kernel(const ArgsFused a, const Taps49 W, const LowRankTaps F) {
    constexpr int IH = 4 * R1;            // intermediate rows
    constexpr int TH = IH - 6;            // output rows
    constexpr int AH = IH + 6;            // input rows
    constexpr int R2 = (TH + 3) / 4;      // output rows per wave (the last wave owns fewer)
    constexpr int BH = 3 * R2 + R2 + 6;   // rows of B the last wave may touch (rows >= IH are never written)
    constexpr int NCHUNK = AH * kInChunks;
    constexpr int NIT = (NCHUNK + 255) / 256;
    // One LDS array: the input window A, then -- once every wave has finished application 1 -- the intermediate
    // tile B written over it from registers.  41 KB per workgroup -> 3 workgroups per CU instead of 2.
    static_assert((BH > IH ? BH : IH) * kMidW <= AH * kInW, "B must fit in A's space");
    __shared__ __attribute__((aligned(16))) double A[AH * kInW];
    double *const B = A;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;

    // tile schedule: one tile per workgroup, or (PERSIST) every `stride`-th tile of this XCD's contiguous run
    const int ntiles = a.tiles_x * a.tiles_y;
    int lin, lin_end, stride;
    if (PERSIST) {
        const int nb = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, slot = b >> 3;
        const int q = ntiles >> 3, rr = ntiles & 7;
        const int start = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
        lin = start + slot;
        lin_end = start + (xcd < rr ? q + 1 : q);
        stride = (nb - xcd + 7) >> 3;  // workgroups on this XCD
    } else {
        lin = xcd_contiguous(blockIdx.x, gridDim.x);
        lin_end = lin + 1;
        stride = 1;
    }
    if (lin >= lin_end) return;

    // ---- staging: interior rows i0-6 .. i0+TH+5, interior columns j0-6 .. j0+129, i.e. padded rows i0-2 ..,
    //      padded columns j0-2 ..; pieces outside the padded array are clamped (they only feed intermediate
    //      cells outside the interior, which are forced to 0 below) -------------------------------------------
    d2 stage[NIT];
    auto fetch = [&](int tile_lin) {
        int fty, ftx;
        panel_major(tile_lin, a.tiles_x, a.tiles_y, a.panel_w, fty, ftx);
        const int fi0 = a.row_begin + fty * TH, fj0 = ftx * kOutW;
        const int max_row = a.m + 7, max_col = a.n + 6;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = tid + it * 256;
            if (NCHUNK % 256 == 0 || k < NCHUNK) {
                const int r = k / kInChunks, c = k - r * kInChunks;
                const int gr = min(max(fi0 - 2 + r, 0), max_row);
                const int gc = min(max(fj0 - 2 + 2 * c, 0), max_col);
                stage[it] = *reinterpret_cast<const d2 *>(a.in + (size_t) gr * a.ld + gc);
            }
        }
    };
    if (!(LORA_ABLATE(a) & 2)) fetch(lin);

    for (; lin < lin_end; lin += stride) {
    int ty, tx;
    panel_major(lin, a.tiles_x, a.tiles_y, a.panel_w, ty, tx);
    const int i0 = a.row_begin + ty * TH;  // first output row (interior coordinates)
    const int j0 = tx * kOutW;             // first output column
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * 256;
        if (NCHUNK % 256 == 0 || k < NCHUNK) *reinterpret_cast<d2 *>(A + 2 * k) = stage[it];
    }
    __syncthreads();
    if (PERSIST && lin + stride < lin_end) fetch(lin + stride);  // in flight during both applications

    // ---- application 1: intermediate rows wv*R1 .. +R1-1, columns 2*lane, 2*lane+1 of B ---------------------
    {
        double acc0[R1], acc1[R1];
#pragma unroll
        for (int r = 0; r < R1; ++r) {
            acc0[r] = 0.0;
            acc1[r] = 0.0;
        }
        const double *strip = A + (wv * R1) * kInW + 2 * lane;  // window = A columns 2*lane .. 2*lane+7
        const int jm = j0 - 3 + 2 * lane;                        // interior column of B column 2*lane
        const bool c0_in = jm >= 0 && jm < a.n;
        const bool c1_in = jm + 1 >= 0 && jm + 1 < a.n;
        d2 cur[4], nxt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const d2 *>(strip + 2 * q);
#pragma unroll
        for (int j = 0; j < R1 + 6; ++j) {
            if (j + 1 < R1 + 6) {
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = *reinterpret_cast<const d2 *>(strip + (j + 1) * kInW + 2 * q);
            }
            double win[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                win[2 * q] = cur[q].x;
                win[2 * q + 1] = cur[q].y;
            }
            apply_row<TAPSET, R1>(j, win, acc0, acc1, W, F);
#pragma unroll
            for (int r = 0; r < R1; ++r) {
                if (j - r >= 0 && j - r < 7) asm volatile("" : "+v"(acc0[r]), "+v"(acc1[r]));
            }
            if (a.dirichlet && j >= 6) {
                // Dirichlet boundary: an intermediate cell outside the interior is a halo cell that keeps the
                // caller's value, which is the cell itself in the input window (3 rows / columns further in A)
                const int r = j - 6;
                const int im = i0 - 3 + wv * R1 + r;
                const bool row_in = im >= 0 && im < a.m;
                const double *cell = A + (wv * R1 + r + 3) * kInW + 2 * lane + 3;
                if (!(row_in && c0_in)) acc0[r] = cell[0];
                if (!(row_in && c1_in)) acc1[r] = cell[1];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // every wave has consumed its part of A: its space now takes the intermediate tile
#pragma unroll
        for (int r = 0; r < R1; ++r) {
            const int im = i0 - 3 + wv * R1 + r;  // interior row of this intermediate row
            const bool row_in = im >= 0 && im < a.m;
            d2 v;
            // cells outside the interior are halo cells of "buffer 1": never written, always 0
            v.x = (a.dirichlet || (row_in && c0_in)) ? acc0[r] : 0.0;
            v.y = (a.dirichlet || (row_in && c1_in)) ? acc1[r] : 0.0;
            *reinterpret_cast<d2 *>(B + (wv * R1 + r) * kMidW + 2 * lane) = v;
        }
    }
    __syncthreads();

    // ---- application 2: output rows wv*R2 .. +R2-1, columns 2*lane, 2*lane+1 (lanes 0..60) -------------------
    {
        double acc0[R2], acc1[R2];
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            acc0[r] = 0.0;
            acc1[r] = 0.0;
        }
        const double *strip = B + (wv * R2) * kMidW + 2 * min(lane, 60);  // window = B columns 2*lane .. 2*lane+7
        const int col = j0 + 2 * lane;
        const bool col_ok = lane < kOutW / 2 && col < a.n;
        d2 cur[4], nxt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const d2 *>(strip + 2 * q);
#pragma unroll
        for (int j = 0; j < R2 + 6; ++j) {
            if (j + 1 < R2 + 6) {
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = *reinterpret_cast<const d2 *>(strip + (j + 1) * kMidW + 2 * q);
            }
            double win[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                win[2 * q] = cur[q].x;
                win[2 * q + 1] = cur[q].y;
            }
            apply_row<TAPSET, R2>(j, win, acc0, acc1, W, F);
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                if (j - r >= 0 && j - r < 7) asm volatile("" : "+v"(acc0[r]), "+v"(acc1[r]));
            }
            if (j >= 6) {
                const int r = j - 6;
                const int ro = wv * R2 + r;  // output row inside the tile
                const int row = i0 + ro;
                if (col_ok && ro < TH && row < a.row_end && !(LORA_ABLATE(a) & 1)) {
                    d2 v;
                    v.x = acc0[r];
                    v.y = acc1[r];
                    *reinterpret_cast<d2 *>(a.out + (size_t) (row + 4) * a.ld + (col + 4)) = v;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (PERSIST) __syncthreads();  // B lives in A's space: the next window may land only after application 2
    }  // tile loop
}

template <int TAPSET, int R1>
hipError_t launch_fused2_t(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    constexpr int TH = 4 * R1 - 6;
    ArgsFused a;
    a.in = in;
    a.out = out;
    a.m = p.dims[0];
    a.n = p.dims[1];
    a.ld = a.n + 8;
    a.row_begin = begin;
    a.row_end = end;
    a.tiles_x = (a.n + kOutW - 1) / kOutW;
    a.tiles_y = (end - begin + TH - 1) / TH;
    a.panel_w = p.panel_width < 1 ? 1 : (p.panel_width > a.tiles_x ? a.tiles_x : p.panel_width);
    a.dirichlet = p.boundary == LORA_BC_DIRICHLET;
    a.ablate = p.ablate;
    Taps49 w;
    for (int k = 0; k < 49; ++k) w.w[k] = p.w[k];
    LowRankTaps f{};
    for (int t = 0; t < 3; ++t)
        for (int e = 0; e < 7; ++e) {
            f.u[t][e] = p.lowrank.u[t][e];
            f.v[t][e] = p.lowrank.v[t][e];
        }
    f.rc = p.lowrank_rc;
    if (TAPSET == EVAL_NEST)
        for (int k = 0; k < 4; ++k) {
            f.u[0][k] = p.nest_g[k];
            f.v[0][k] = p.nest_a[k];
        }
    const long nblocks = (long) a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    if (p.persistent) {
        // 3 resident workgroups per CU (LDS-limited), a multiple of 8 so that every XCD gets the same count
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        long grid = 3L * cus;
        if (grid > nblocks) grid = nblocks;
        hipLaunchKernelGGL((stencil2d_fused2_kernel<TAPSET, R1, true>), dim3((unsigned) grid), dim3(256), 0, s, a, w, f);
    } else {
        hipLaunchKernelGGL((stencil2d_fused2_kernel<TAPSET, R1, false>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w, f);
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_2d_fused2(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
#define LORA_FUSED_DISPATCH(R1)                                                        \
    if (p.fused_eval == EVAL_NEST) return launch_fused2_t<EVAL_NEST, R1>(p, in, out, begin, end, s); \
    if (p.fused_eval == EVAL_LR_DIAMOND) return launch_fused2_t<EVAL_LR_DIAMOND, R1>(p, in, out, begin, end, s); \
    if (p.fused_eval == EVAL_LR_PYRAMID) return launch_fused2_t<EVAL_LR_PYRAMID, R1>(p, in, out, begin, end, s); \
    if (p.fused_eval == EVAL_LR_PYRAMID_SYM) return launch_fused2_t<EVAL_LR_PYRAMID_SYM, R1>(p, in, out, begin, end, s); \
    if (p.fused_eval == EVAL_LR_PYRAMID_SYM_GAP) return launch_fused2_t<EVAL_LR_PYRAMID_SYM_GAP, R1>(p, in, out, begin, end, s); \
    switch (p.tapset) {                                                                 \
        case TAPS2D_DIAMOND:                                                            \
            return launch_fused2_t<TAPS2D_DIAMOND, R1>(p, in, out, begin, end, s);     \
        case TAPS2D_STAR:                                                               \
            return launch_fused2_t<TAPS2D_STAR, R1>(p, in, out, begin, end, s);        \
        default:                                                                        \
            return launch_fused2_t<TAPS2D_BOX, R1>(p, in, out, begin, end, s);         \
    }
    if (p.fused_rows == 6) { LORA_FUSED_DISPATCH(6) }
    if (p.fused_rows == 10) { LORA_FUSED_DISPATCH(10) }
    LORA_FUSED_DISPATCH(8)
#undef LORA_FUSED_DISPATCH
}

const char *kernel_name_2d_fused2(const Plan &) { return "stencil2d_fused2_kernel"; }

}  // namespace lora
