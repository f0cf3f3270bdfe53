// kernels_2d.hip -- 2D sweeps for gfx950 (MI355X): one kernel application of a radius-3 stencil on a
// padded (m+8) x (n+8) fp64 grid.  Replaces kernel2d_{star2d1r,star2d3r,box2d3r} of the reference
// (2d/gpu.cu:31-273); written for 64-wide wavefronts, not a translation of the wmma tiling.
//
// DIRECT variant (this file, stencil2d_direct_kernel)
//   * One 256-thread workgroup owns a TH x 128 output tile (TH = 4 waves x RPT rows).  The
//     (TH+6) x 136 input window -- the tile plus its halo ring, widened to 16-byte alignment -- is
//     fetched with 16-byte coalesced loads (a wave reads ~one 1 KiB row segment per instruction) and
//     staged in LDS once.
//   * Each lane owns two adjacent columns and RPT rows.  It walks the RPT+6 input rows of its strip;
//     per row it reads a 10-wide window from LDS (5 x ds_read_b128, conflict-free: consecutive lanes
//     read consecutive 16-byte slots) and scatters it into the RPT x 2 register accumulators of the
//     output rows that row contributes to.  That is 25 / 13 / 49 v_fma_f64 per point (diamond / star /
//     box tap set) and (RPT+6)*5/(2*RPT) LDS reads per point instead of one per tap.
//   * Taps are applied in row-major tap order (dy, then dx), i.e. the summation order of the reference's
//     CPU check (2d/main.cu:38-93), fused multiply-add instead of multiply + add.
//   * Block -> tile map: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), so each XCD gets a
//     contiguous run of tiles in a panel-major order (panels of `panel_width` tile columns walked top
//     to bottom).  Horizontal halo columns and the 6 shared halo rows of vertically adjacent tiles are
//     then re-read from that XCD's L2 instead of HBM.
//   * Stores: each lane writes 16 bytes, a wave one contiguous 1 KiB row segment; halo cells are never
//     written (2d/gpu.cu:266-271).
//   * Partial tiles are guarded (loads clamped into the padded array, stores predicated), so any m and
//     any even n work; the reference has no guards.
#include <hip/hip_runtime.h>

#include "device_common.h"

namespace lora {

namespace {

constexpr int kTileW = 128;            // output columns per tile: 64 lanes x 2
constexpr int kLdsW = kTileW + 8;      // staged columns (halo 3 each side, widened to 4 for alignment)
constexpr int kChunksPerRow = kLdsW / 2;  // 16-byte chunks per staged row

struct Args2D {
    const double *in;
    double *out;
    int ld;         // padded row length n + 8
    int m, n;       // interior extents
    int row_begin;  // first interior row of this launch (multiple of the tile height)
    int row_end;    // one past the last interior row of this launch
    int tiles_x, tiles_y;
    int panel_w;
};

template <int TAPSET, int RPT, bool NT>
__global__ __launch_bounds__(256, (RPT <= 4 ? 6 : (RPT <= 8 ? 3 : 2))) void stencil2d_direct_kernel(const Args2D a, const Taps49 W) {
    constexpr int TH = 4 * RPT;
    constexpr int LH = TH + 6;
    constexpr int NCHUNK = LH * kChunksPerRow;
    constexpr int NIT = (NCHUNK + 255) / 256;
    __shared__ __attribute__((aligned(16))) double tile[LH * kLdsW];

    const int tid = threadIdx.x;
    int ty, tx;
    panel_major(xcd_contiguous(blockIdx.x, gridDim.x), a.tiles_x, a.tiles_y, a.panel_w, ty, tx);
    const int i0 = a.row_begin + ty * TH;  // first interior row of the tile
    const int j0 = tx * kTileW;            // first interior column of the tile

    // ---- stage the input window: padded rows i0+1 .. i0+TH+6, padded columns j0 .. j0+135 ------
    {
        d2 stage[NIT];
        const int max_row = a.m + 7;  // last padded row
        const int max_col = a.n + 6;  // last 16-byte chunk start in a padded row
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = tid + it * 256;
            if (NCHUNK % 256 == 0 || k < NCHUNK) {
                const int r = k / kChunksPerRow;
                const int c = k - r * kChunksPerRow;
                const int gr = min(i0 + 1 + r, max_row);
                const int gc = min(j0 + 2 * c, max_col);
                stage[it] = *reinterpret_cast<const d2 *>(a.in + (size_t) gr * a.ld + gc);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = tid + it * 256;
            if (NCHUNK % 256 == 0 || k < NCHUNK) *reinterpret_cast<d2 *>(tile + 2 * k) = stage[it];
        }
    }
    __syncthreads();

    // ---- compute: lane owns tile columns 2*lane+4, 2*lane+5 (window 2*lane .. 2*lane+9) ----------
    const int lane = tid & 63;
    const int wv = tid >> 6;
    double acc0[RPT], acc1[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        acc0[r] = 0.0;
        acc1[r] = 0.0;
    }
    const double *strip = tile + (wv * RPT) * kLdsW + 2 * lane;
    const int col = j0 + 2 * lane;
    // The window of row j+1 is fetched while row j is being consumed (two register sets); the
    // sched_barrier keeps the compiler from hoisting every row's reads to the top, which would
    // blow the register budget the launch bounds grant.
    d2 cur[5], nxt[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) cur[q] = *reinterpret_cast<const d2 *>(strip + 2 * q);
#pragma unroll
    for (int j = 0; j < RPT + 6; ++j) {
        if (j + 1 < RPT + 6) {
#pragma unroll
            for (int q = 0; q < 5; ++q) nxt[q] = *reinterpret_cast<const d2 *>(strip + (j + 1) * kLdsW + 2 * q);
        }
        double win[10];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            win[2 * q] = cur[q].x;
            win[2 * q + 1] = cur[q].y;
        }
        // input row j of the strip is tap row dy = j - r of output row r
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int dy = j - r;
            if (dy >= 0 && dy < 7) {
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    if (tap_on<TAPSET>(dy, dx)) {
                        const double wt = W.w[dy * 7 + dx];
                        acc0[r] = fma(wt, win[dx + 1], acc0[r]);
                        acc1[r] = fma(wt, win[dx + 2], acc1[r]);
                    }
                }
            }
        }
        // Pin the partial sums here: without this the optimiser sinks every FMA chain into the
        // predicated store block of its row, which keeps all RPT+6 windows alive at once.
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (j - r >= 0 && j - r < 7) asm volatile("" : "+v"(acc0[r]), "+v"(acc1[r]));
        }
        // output row j-6 is complete: store it (16 bytes per lane, 1 KiB contiguous per wave); halo
        // cells are never written (2d/gpu.cu:266-271)
        if (j >= 6) {
            const int r = j - 6;
            const int row = i0 + wv * RPT + r;
            if (col < a.n && row < a.row_end) {
                d2 v;
                v.x = acc0[r];
                v.y = acc1[r];
                d2 *dstp = reinterpret_cast<d2 *>(a.out + (size_t) (row + 4) * a.ld + (col + 4));
                if (NT)
                    __builtin_nontemporal_store(v, dstp);  // written once, read next launch: keep L2 for the halos
                else
                    *dstp = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) cur[q] = nxt[q];
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int TAPSET, int RPT, bool NT>
hipError_t launch_direct(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    constexpr int TH = 4 * RPT;
    Args2D a;
    a.in = in;
    a.out = out;
    a.m = p.dims[0];
    a.n = p.dims[1];
    a.ld = a.n + 8;
    a.row_begin = begin;
    a.row_end = end;
    a.tiles_x = (a.n + kTileW - 1) / kTileW;
    a.tiles_y = (end - begin + TH - 1) / TH;
    a.panel_w = p.panel_width < 1 ? 1 : (p.panel_width > a.tiles_x ? a.tiles_x : p.panel_width);
    Taps49 w;
    for (int k = 0; k < 49; ++k) w.w[k] = p.w[k];
    const long nblocks = (long) a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((stencil2d_direct_kernel<TAPSET, RPT, NT>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    return hipGetLastError();
}

template <int TAPSET>
hipError_t launch_direct_rpt(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (p.nt_store) {
        switch (p.rows_per_thread) {
            case 4:
                return launch_direct<TAPSET, 4, true>(p, in, out, begin, end, s);
            case 16:
                return launch_direct<TAPSET, 16, true>(p, in, out, begin, end, s);
            default:
                return launch_direct<TAPSET, 8, true>(p, in, out, begin, end, s);
        }
    }
    switch (p.rows_per_thread) {
        case 4:
            return launch_direct<TAPSET, 4, false>(p, in, out, begin, end, s);
        case 16:
            return launch_direct<TAPSET, 16, false>(p, in, out, begin, end, s);
        default:
            return launch_direct<TAPSET, 8, false>(p, in, out, begin, end, s);
    }
}

}  // namespace

hipError_t launch_2d_direct(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    switch (p.tapset) {
        case TAPS2D_DIAMOND:
            return launch_direct_rpt<TAPS2D_DIAMOND>(p, in, out, begin, end, s);
        case TAPS2D_STAR:
            return launch_direct_rpt<TAPS2D_STAR>(p, in, out, begin, end, s);
        default:
            return launch_direct_rpt<TAPS2D_BOX>(p, in, out, begin, end, s);
    }
}

const char *kernel_name_2d_direct(const Plan &) { return "stencil2d_direct_kernel"; }

}  // namespace lora
