// kernels_3d_lanes.hip -- FOUR applications of a 3D radius-1 stencil per launch (fp64), the time levels held in registers.
//
// Where round 2 left 3D (profiles/r02_star3d1r_pmc.json): the plane-streaming kernel of
// kernels_3d_planes.hip keeps every level as a 66 x 68 tile in LDS.  Four tiles are 142 KB, so it stops at three
// applications per launch and one 8-wave workgroup per CU; per level a wave reads 18 x 16-byte windows and writes 4 to
// compute 56 multiply-adds, and with K barriers per plane all eight waves read, then compute, then write -- the LDS pipe
// (992 cycles per level and CU, 42 % of them the 79 B/clk ds_write_b128 path) and the vector pipe (448) take turns:
// ~6500 cycles per plane step where neither pipe needs 3000.  And at three applications per launch with 1.2 x read
// amplification the launch would hit the HBM ceiling near 750 GStencils/s anyway.
//
// Here a level never goes to LDS as a tile:
//   * A lane owns the same 4 rows x 2 columns of a (4 NW) x 128 tile at EVERY level.  x-neighbours are the neighbouring
//     lanes' registers (v_mov_b32_dpp wave_shr:1 / wave_shl:1: 4 moves per row and level, each at the issue cost of one
//     fp64 multiply-add -- tools/probes/dpp_rate_probe.hip); y-neighbours are the lane's own rows, and across waves the
//     two edge rows each wave publishes per level (2 ds_write_b128 + 2 ds_read_b128 per wave and level instead of 4 + 18);
//     z is streamed: per level and point two partial sums live between steps (the plane above has its dz = 0 tap, the
//     plane at hand its dz = 0 and dz = 1 taps), the third is completed in the step and becomes the next level's input.
//   * So K = 4 fits: 4 x 32 partial-sum registers + planes in flight = 252 VGPRs, two waves per SIMD; LDS: 64 KB of edge
//     rows + the 96 KB delay line of the EDGE steps = the CU's whole 160 KB (K = 2: 32 KB).  Registers AND LDS hold it to
//     one 512-thread workgroup per CU.
//     The grid is read once and written once per FOUR sweeps (4 B per point and sweep compulsory).
//   * Per accumulator the taps arrive in the oracle's order (dz, then dy, then dx): bit-identical to four single sweeps
//     (and to the other fused 3D kernels), also for the separable box (x-pass, y-pass, z-scatter as in planes_3d.h).
//   * Input planes come straight into registers (4 x global_load_dwordx4 per lane and plane), one plane ahead; the plane
//     completed in a step is stored at the START of the next one, right behind the step's single s_waitcnt vmcnt(0) --
//     everything that wait covers (the loads of this step's plane, the stores of the plane before) was issued a whole
//     step earlier.
//   * Halo semantics as in 2D: cells of an intermediate level outside the interior are 0 at odd levels and the
//     source buffer's own halo value at even ones (level 2).  Tiles on the grid's rim, and every tile in the few steps
//     whose planes lie outside the z range, run an EDGE copy of the step that forces them.  The level-2 value of a cell is
//     the input's value of the same cell two planes back in the stream -- a cell the lane itself loaded two steps earlier:
//     it comes out of a private three-plane delay line in LDS (4 ds_write_b128 + 4 ds_read_b128 per step; fetching it
//     again by LDS-DMA cost four ~100-cycle instruction issues per step and made rim tiles a third slower).
//   * One round: tiles x z-chunks = the workgroups resident at once (one per CU), chunks as long as that allows.
//   * Odd innermost extents: rows are then only 8-byte aligned, which 16-byte global loads tolerate; the stores go through
//     a buffer descriptor whose range check cuts the half-valid last pair (the 2D kernels' trick).
//
// Taps: the 7-point star (any weights) and exactly separable 27-point boxes (the reference's); other 27-tap tables stay
// with kernels_3d_planes.hip / kernels_3d_fused.hip.  Replaces the reference's z-loop 3d/gpu_star.cu:101-133 /
// gpu_box.cu:105-140 and time-step loop 3d/gpu_star.cu:177-181 (four steps per pass).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "device_common.h"
#include "spans.h"

// Ablation switches for tools/probes/lanes3_ablate.hip (timing only -- results are wrong with any of them set):
// 1 no barriers, 2 no stores, 4 no plane loads after the first, 8 no cross-lane moves, 16 no EDGE steps, 32 only EDGE steps,
// 64 (star) only the chain of dz = 2 taps: the memory pattern with next to no arithmetic
#ifndef LORA_L3_ABLATE
#define LORA_L3_ABLATE 0
#endif

namespace lora {

namespace {

constexpr int kTileW = 128;  // columns of a tile: 64 lanes x 2
constexpr int kOutW = 120;   // output columns of a tile: lanes 2 .. 61 (K <= 4 columns lost per side, pairs kept whole)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ArgsL3 {
    const double *in;
    double *out;
    int h, m, n;
    int ld;
    long plane;
    int z_begin, z_end;
    int z_begin2, z_end2;  // a second range of planes in the same launch (chunks only): the two end regions of a slab
    int zc;
    int tiles_x, tiles_y;
#ifdef LORA_L3_STAMP
    long long *stamps;  // probe builds: per workgroup {start, end} on the 100 MHz clock, {tile x | y << 16 | rim << 31, first plane}
#endif
    int deal;  // chunks, more workgroups' worth than slots: the launch is `slots` workgroups and chunk j goes to workgroup j mod slots (0: off)
    int slow_c, slow_zc;  // chunks, one round with slots to spare: the tiles of the first tile column and row get slow_c chunks of slow_zc planes (0: off)
    Spans sp;  // zc == 0: spans (spans.h)
    int team;  // zc == 0: 0 = a span per workgroup over all tiles; TX = a span per TEAM of the TX workgroups of a tile row
};

__device__ __forceinline__ double lane_below(double v) {  // the value lane i - 1 holds (0 in lane 0)
    if constexpr (LORA_L3_ABLATE & 8) return v;
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) b, 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned) lo);
}
__device__ __forceinline__ double lane_above(double v) {  // the value lane i + 1 holds (0 in lane 63)
    if constexpr (LORA_L3_ABLATE & 8) return v;
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) b, 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned) lo);
}

// Step p of a chunk whose first output plane is k0 takes input plane k0 - K + p; level l completes its plane
// k0 - K + p - l in that step (one plane behind the level below); level K is the output, stored in step p + 1.
template <int TAPSET, int K, int NW>
__global__ __launch_bounds__(NW * 64, 2) void stencil3d_lanes_kernel(const ArgsL3 a, const Taps27 W) {
    constexpr int OH = 4 * NW - 2 * K;  // output rows of a tile
    static_assert(K == 2 || K == 4, "an even number of applications (fused launches start at even steps)");
    static_assert(TAPSET == TAPS3D_STAR || TAPSET == TAPS3D_SEP, "star or separable box");
    __shared__ __attribute__((aligned(16))) double edge_rows[K][NW][2][kTileW];  // level l: each wave's first / last row
    __shared__ __attribute__((aligned(16))) double delay[K > 2 ? 3 : 1][NW][4][kTileW];  // EDGE: this wave's input planes of the last three steps

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int TX = a.tiles_x, TY = a.tiles_y;
    // Spans (a.zc == 0, spans.h): the workgroup owns a range of the line of all (tile, plane) pairs and runs one SEGMENT per
    // tile the range touches.  Chunks: one segment, chunk `lin / tiles` of its tile -- the tiles on the rim of the grid first
    // (they run the slower EDGE steps throughout, and with the long workgroups dispatched first the short ones even out
    // the end of the launch).
    unsigned v0 = 0, v1 = 0;
    if (a.zc == 0) span_range(a.sp, a.team ? lin / a.team : lin, v0, v1);
    int job = lin;  // chunks dealt out (a.deal): this workgroup's next chunk
    for (bool more = true, first = true; more; first = false) {
    int tx, ty, k0, zc;
    if (a.zc == 0) {
        int z0;
        bool any;
        if (a.team) {  // the line runs over tile ROWS; this workgroup is column lin mod TX of its team's row
            int t0;
            any = span_next(a.sp, 2 * K + 1, v0, v1, 1, TY, t0, ty, z0, zc);
            tx = lin - (lin / a.team) * a.team;
        } else {
            any = span_next(a.sp, 2 * K + 1, v0, v1, TX, TY, tx, ty, z0, zc);
        }
        more = v0 < v1;
        if (!any) continue;
        k0 = a.z_begin + z0;
    } else {
        int chunk;
        const int c0 = (a.z_end - a.z_begin + a.zc - 1) / a.zc, c1 = (a.z_end2 - a.z_begin2 + a.zc - 1) / a.zc;
        if (a.slow_c > 0) {
            // A launch of ONE round with slots to spare: the full rim tiles -- first tile column and first tile row: EDGE
            // steps on every lane's worth of memory traffic, 8 - 17 % longer than an inner tile's (tools/probes/
            // lanes3_timeline.hip) -- are cut into one chunk more than the others, so that the launch no longer waits for them.
            const int nslow = TX + TY - 1, na = nslow * a.slow_c;
            if (lin < na) {
                const int j = lin / a.slow_c;
                chunk = lin - j * a.slow_c;
                tx = j < TY ? 0 : j - TY + 1;
                ty = j < TY ? j : 0;
                k0 = a.z_begin + chunk * a.slow_zc;
                zc = min(a.slow_zc, a.z_end - k0);
            } else {
                const int l2 = lin - na, i = l2 / c0;
                chunk = l2 - i * c0;
                ty = 1 + i / (TX - 1);
                tx = 1 + i - (ty - 1) * (TX - 1);
                k0 = a.z_begin + chunk * a.zc;
                zc = min(a.zc, a.z_end - k0);
            }
        } else {
            // (a.deal: the chunks of a launch of several rounds are DEALT to the resident workgroups, chunk j to workgroup
            // j mod slots, instead of being dispatched one workgroup each as slots fall free.  Every workgroup then runs the
            // same number of chunks and about the same number of rim chunks.  Dispatched one by one, the 1792 chunks of
            // box3d1r 768^3 -- seven rounds exactly, rim chunks 259 us, inner ones 225 -- left some CUs with eight chunks and
            // others with six: the last tenth of the launch ran on 96 of 256 CUs, profiles/r04_lanes3_timeline.txt.)
            chunk_of(job, c0 + c1, TX, TY, chunk, tx, ty);
            const int zb = chunk < c0 ? a.z_begin : a.z_begin2, ze = chunk < c0 ? a.z_end : a.z_end2;
            k0 = zb + (chunk < c0 ? chunk : chunk - c0) * a.zc;
            zc = min(a.zc, ze - k0);
            job += (int) gridDim.x;
        }
        more = a.deal > 0 && job < a.deal;
    }
    // (the segment before is done with the rows in LDS when its slowest wave is)
    if (!first) __builtin_amdgcn_s_barrier();
#ifdef LORA_L3_STAMP
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        a.stamps[4 * blockIdx.x + 3] = k0;
    }
#endif
    const int X0 = tx * kOutW - 4, Y0 = ty * OH - K;  // interior coordinates of the tile's first column / row
    // this lane's cells: rows Y0 + 4 wv + r (r = 0 .. 3), columns X0 + 2 lane, + 1; padded: + 2 rows, + 4 columns, clamped
    // into the padded array (clamped cells only feed cells outside the interior, which EDGE forces, or nothing)
    const int col = X0 + 2 * lane;
    const int pc = min(max(col + 4, 0), a.n + 6);
    long rowoff[4];  // uniform: padded row x ld
#pragma unroll
    for (int r = 0; r < 4; ++r) rowoff[r] = (long) min(max(Y0 + 4 * wv + r + 2, 0), a.m + 3) * a.ld;
    const int up = max(wv - 1, 0), dn = min(wv + 1, NW - 1);  // (the tile's outermost rows are never valid beyond level 0)
    // (An L2 prefetch of the planes two or three steps ahead -- one 4-byte LDS-DMA per lane and step touching the wave's
    // 64-byte pieces, landing in an LDS row nobody reads -- made the launch 5-12 % SLOWER, with and without the arithmetic:
    // the launch is bound by the bytes this access pattern moves, 2.7 GB at ~4.9 TB/s, not by the latency of its loads.
    // tools/probes/lanes3_ablate.hip, profiles/r03_lanes3d_ablation.txt.)

    // stores: lanes 2 .. 61 write interior columns col, col + 1 of rows K .. 4 NW - K - 1 of the tile; the descriptor's
    // range check drops what lies beyond the row (also the second half of the last pair when n is odd)
    const unsigned st_off = (lane >= 2 && lane < 62 && col >= 0) ? 8u * (unsigned) col : 0x80000000u;
    bool st_row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int trow = 4 * wv + r;
        st_row[r] = trow >= K && trow < 4 * NW - K && Y0 + trow < a.m;
    }

    // EDGE: which of this lane's cells are interior cells in x and y
    bool in0[4], in1[4];
    const bool xy_rim = X0 < 0 || X0 + kTileW > a.n || Y0 < 0 || Y0 + 4 * NW > a.m;  // (uniform over the workgroup)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = Y0 + 4 * wv + r;
        const bool row_in = (unsigned) row < (unsigned) a.m;
        in0[r] = row_in && (unsigned) col < (unsigned) a.n;
        in1[r] = row_in && (unsigned) (col + 1) < (unsigned) a.n;
    }

    // Per level and point three sums rotate through three register slots, one step (= plane) per turn; in a step of
    // phase P (= p mod 3, a compile-time constant of the step's copy):
    //   acc[l][P]        the plane at hand (dz = 0, 1 taps in); the dz = 2 tap completes it IN PLACE, and it is then the
    //                    input plane of level l + 1 for the rest of the step (level K: the output, stored next step)
    //   acc[l][(P+1)%3]  the plane above (dz = 0 tap in); takes its dz = 1 taps in place
    //   acc[l][(P+2)%3]  free (last step's completed plane, consumed by then): opened with the dz = 0 tap
    // With the roles turning instead of the values no register is ever copied (the two-slot form of this loop spent a fifth
    // of its vector instructions on v_mov_b64).
    double acc[K][3][4][2];
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[l][q][r][0] = acc[l][q][r][1] = 0.0;
    d2 nxt[3][4];  // input planes in flight: plane p in slot p mod 3 (two are live at a time)
    auto load_plane = [&](int p, d2 (&dst)[4]) {
        const double *src = a.in + (long) min(max(k0 - K + p + 1, 0), a.h + 1) * a.plane + pc;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r] = *reinterpret_cast<const d2 *>(src + rowoff[r]);  // (compiler-tracked: see step())
    };
    load_plane(0, nxt[0]);

    auto step = [&](const int p, auto phase_tag, auto edge_tag, auto levels_tag) {
        constexpr int P = decltype(phase_tag)::value, P1 = (P + 1) % 3, P2 = (P + 2) % 3;
        constexpr bool EDGE = decltype(edge_tag)::value;
        // NA < K: a copy of the step for a segment's first steps, where the upper levels have nothing to do yet.  Level
        // index l (levels l -> l + 1) takes in plane k0 - K + p - l, which feeds planes ... - 1 .. + 1 of level l + 1; the
        // segment needs that level's planes from k0 - (K - l - 1) on: the level has work from step 2 l.  Steps 0 .. 2 K - 3
        // run min(K, p / 2 + 1) levels: 12 level-steps = 3 steps' worth of a segment's zc + 2 K + 1 (K = 4) -- 3 % of the
        // 96-plane chunks of box3d1r 768^3, 4 % of a 64-plane slab.
        constexpr int NA = decltype(levels_tag)::value;
        // Everything issued one step ago has had a whole step to complete: the loads of plane p, the stores of the output
        // plane before last.  The plane loads are plain loads on purpose: the compiler waits for them where they are first
        // used -- right here, and with stores outstanding beside them it waits for vmcnt(0), which is what this step wants
        // anyway -- and it never copies or spills a register whose load is still in flight (loads issued and waited for in
        // separate asm statements gave wrong planes at 256 VGPRs).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            const int o = p - 2 * K - 1;  // the output plane completed in the previous step: level K's slot of phase P - 1
            const bool live = o >= 0 && o < zc;
            double *const dst = a.out + (long) (k0 + max(o, 0) + 1) * a.plane + 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    dst + rowoff[r], 0, (live && st_row[r]) ? (unsigned) a.n * 8u : 0u, 0x00020000);
                const d2 ov = {acc[K - 1][P2][r][0], acc[K - 1][P2][r][1]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), rs, (LORA_L3_ABLATE & 2) ? 0x80000000u : st_off, 0, kStoreNT);
            }
        }
        double v0[4][2];  // the input plane of level 1
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v0[r][0] = nxt[P][r].x;
            v0[r][1] = nxt[P][r].y;
        }
        if constexpr (LORA_L3_ABLATE & 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) nxt[P1][r] = nxt[P][r] + (d2){1e-9, 1e-9};
        } else {
            load_plane(p + 1, nxt[P1]);
        }
        if (EDGE && K > 2) {
            // The value a level-2 cell outside the interior is forced to is the source buffer's own value there (while fused
            // launches run every buffer carries buffer 0's halo) -- the very cell this lane holds of the INPUT plane with the
            // same index, which it took in two steps ago: a private delay line of three planes in LDS, no memory request.
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<d2 *>(&delay[P][wv][r][2 * lane]) = (d2){v0[r][0], v0[r][1]};
        }
        // the plane of level l + 1 just completed, k0 - K + p - (l + 1): its cells outside the interior are forced
        auto force = [&](const int l) {
            if (EDGE && l + 1 < K) {
                const int z = k0 - K + p - (l + 1);
                const bool z_in = (unsigned) z < (unsigned) a.h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d2 hv = {0.0, 0.0};
                    if ((l + 1) % 2 == 0) hv = *reinterpret_cast<const d2 *>(&delay[P1][wv][r][2 * lane]);  // written in step p - 2
                    acc[l][P][r][0] = (z_in && in0[r]) ? acc[l][P][r][0] : hv.x;
                    acc[l][P][r][1] = (z_in && in1[r]) ? acc[l][P][r][1] : hv.y;
                }
            }
        };
        if constexpr (TAPSET == TAPS3D_STAR) {
            // The star's dz = 2 tap is the centre point alone: the planes of ALL levels complete in a chain of K x 8
            // multiply-adds with no neighbour in it, before any exchange.  So the step has ONE exchange: every level's two
            // edge rows are published together, and while they travel the wave does everything that needs no other wave's
            // row (rows 1 .. 3 of the lane's four, but for row 3's last tap); behind the barrier come row 0 and that tap.
            // A second barrier right before the publication keeps it behind the last step's reads.
#pragma unroll
            for (int l = 0; l < NA; ++l) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[l][P][r][0] = fma(W.w[22], l == 0 ? v0[r][0] : acc[l - 1][P][r][0], acc[l][P][r][0]);
                    acc[l][P][r][1] = fma(W.w[22], l == 0 ? v0[r][1] : acc[l - 1][P][r][1], acc[l][P][r][1]);
                }
                force(l);
            }
            // nobody publishes this step's rows before everybody has read the last step's
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(LORA_L3_ABLATE & 1)) asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int l = 0; l < NA; ++l) {
                const double(&in)[4][2] = l == 0 ? v0 : acc[l - 1][P];
                *reinterpret_cast<d2 *>(&edge_rows[l][wv][0][2 * lane]) = (d2){in[0][0], in[0][1]};
                *reinterpret_cast<d2 *>(&edge_rows[l][wv][1][2 * lane]) = (d2){in[3][0], in[3][1]};
            }
            __builtin_amdgcn_sched_barrier(0);
            // taps in the oracle's order per accumulator: (dz = 1) dy = -1, dx = -1, 0, +1, dy = +1; then dz = 0 opens the next
            auto own_rows = [&](const int l) {
                if constexpr (LORA_L3_ABLATE & 64) return;
                const double(&in)[4][2] = l == 0 ? v0 : acc[l - 1][P];
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    const double xl = lane_below(in[r][1]), xr = lane_above(in[r][0]);
                    double s0 = fma(W.w[10], in[r - 1][0], acc[l][P1][r][0]), s1 = fma(W.w[10], in[r - 1][1], acc[l][P1][r][1]);
                    s0 = fma(W.w[12], xl, s0);
                    s1 = fma(W.w[12], in[r][0], s1);
                    s0 = fma(W.w[13], in[r][0], s0);
                    s1 = fma(W.w[13], in[r][1], s1);
                    s0 = fma(W.w[14], in[r][1], s0);
                    s1 = fma(W.w[14], xr, s1);
                    if (r < 3) {
                        s0 = fma(W.w[16], in[r + 1][0], s0);
                        s1 = fma(W.w[16], in[r + 1][1], s1);
                    }
                    acc[l][P1][r][0] = s0;
                    acc[l][P1][r][1] = s1;
                    acc[l][P2][r][0] = fma(W.w[4], in[r][0], 0.0);
                    acc[l][P2][r][1] = fma(W.w[4], in[r][1], 0.0);
                }
            };
#pragma unroll
            for (int l = 0; l + 1 < NA; ++l) own_rows(l);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(LORA_L3_ABLATE & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            d2 vu[K], vd[K];
#pragma unroll
            for (int l = 0; l < NA; ++l) {
                vu[l] = *reinterpret_cast<const d2 *>(&edge_rows[l][up][1][2 * lane]);
                vd[l] = *reinterpret_cast<const d2 *>(&edge_rows[l][dn][0][2 * lane]);
            }
            __builtin_amdgcn_sched_barrier(0);
            own_rows(NA - 1);  // (while the neighbours' rows arrive)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < ((LORA_L3_ABLATE & 64) ? 0 : NA); ++l) {
                const double(&in)[4][2] = l == 0 ? v0 : acc[l - 1][P];
                const double xl = lane_below(in[0][1]), xr = lane_above(in[0][0]);
                double s0 = fma(W.w[10], vu[l].x, acc[l][P1][0][0]), s1 = fma(W.w[10], vu[l].y, acc[l][P1][0][1]);
                s0 = fma(W.w[12], xl, s0);
                s1 = fma(W.w[12], in[0][0], s1);
                s0 = fma(W.w[13], in[0][0], s0);
                s1 = fma(W.w[13], in[0][1], s1);
                s0 = fma(W.w[14], in[0][1], s0);
                s1 = fma(W.w[14], xr, s1);
                acc[l][P1][0][0] = fma(W.w[16], in[1][0], s0);
                acc[l][P1][0][1] = fma(W.w[16], in[1][1], s1);
                acc[l][P2][0][0] = fma(W.w[4], in[0][0], 0.0);
                acc[l][P2][0][1] = fma(W.w[4], in[0][1], 0.0);
                acc[l][P1][3][0] = fma(W.w[16], vd[l].x, acc[l][P1][3][0]);
                acc[l][P1][3][1] = fma(W.w[16], vd[l].y, acc[l][P1][3][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // separable taps w = a(z) b(y) c(x), W.w[0..2] = c, [3..5] = b, [6..8] = a (planes_3d.h, scatter_plane_sep):
            // x-pass, y-pass, then the three z contributions.  The dz = 2 contribution needs the whole (x, y) neighbourhood,
            // so here the levels do follow each other through an exchange each.  What a wave publishes is the x-PASSED
            // form of its first and last row (the same multiply-adds on the same cells whoever does them): the neighbours'
            // rows then cost no x-pass of their own, and the wave's own x-pass runs ahead of the barrier.
#pragma unroll
            for (int l = 0; l < NA; ++l) {
                const double(&in)[4][2] = l == 0 ? v0 : acc[l - 1][P];
                double t0[6], t1[6];
#pragma unroll
                for (int j = 1; j < 5; ++j) {
                    const double c0 = in[j - 1][0], c1 = in[j - 1][1];
                    const double xl = lane_below(c1), xr = lane_above(c0);
                    t0[j] = fma(W.w[2], c1, fma(W.w[1], c0, W.w[0] * xl));
                    t1[j] = fma(W.w[2], xr, fma(W.w[1], c1, W.w[0] * c0));
                }
                *reinterpret_cast<d2 *>(&edge_rows[l][wv][0][2 * lane]) = (d2){t0[1], t1[1]};
                *reinterpret_cast<d2 *>(&edge_rows[l][wv][1][2 * lane]) = (d2){t0[4], t1[4]};
                if constexpr (!(LORA_L3_ABLATE & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const d2 vu = *reinterpret_cast<const d2 *>(&edge_rows[l][up][1][2 * lane]);
                const d2 vd = *reinterpret_cast<const d2 *>(&edge_rows[l][dn][0][2 * lane]);
                t0[0] = vu.x;
                t1[0] = vu.y;
                t0[5] = vd.x;
                t1[5] = vd.y;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double u0 = fma(W.w[5], t0[r + 2], fma(W.w[4], t0[r + 1], W.w[3] * t0[r]));
                    const double u1 = fma(W.w[5], t1[r + 2], fma(W.w[4], t1[r + 1], W.w[3] * t1[r]));
                    acc[l][P][r][0] = fma(W.w[8], u0, acc[l][P][r][0]);
                    acc[l][P][r][1] = fma(W.w[8], u1, acc[l][P][r][1]);
                    acc[l][P1][r][0] = fma(W.w[7], u0, acc[l][P1][r][0]);
                    acc[l][P1][r][1] = fma(W.w[7], u1, acc[l][P1][r][1]);
                    acc[l][P2][r][0] = W.w[6] * u0;
                    acc[l][P2][r][1] = W.w[6] * u1;
                }
                force(l);
            }
            // A level's edge rows are rewritten one step later, and what keeps that behind the neighbours' reads of this step
            // is the barrier of ANOTHER level in between.  A step that runs one level alone has none: it closes with one of
            // its own.  (Without it a wave that was a whole x-pass ahead replaced its row under a neighbour's read -- seen
            // as errors of 1e-3 in five of 600 random box runs, never twice in the same run: tools/fuzz_r04.py.)
            if constexpr (NA == 1 && !(LORA_L3_ABLATE & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };

    // Steps whose intermediate planes can lie outside the z range: p < p1 (below plane 0) and p >= p2 (beyond plane h - 1;
    // two steps early, because the delay line must hold the planes an EDGE step looks back at).  Tiles on the rim in x / y
    // run the EDGE copy throughout.  Steps come in turns of three (one per phase); the EDGE ranges are widened to whole
    // turns, and a chunk runs up to two steps past its last plane (loads clamped, stores switched off).
    const int steps = zc + 2 * K + 1, turns = (steps + 2) / 3;
    int p1 = min(max(2 * K - k0, 0), steps), p2 = min(max(a.h - k0 + K - 1, p1), steps);
    if (xy_rim) p1 = steps;
    if (LORA_L3_ABLATE & 16) p1 = 0, p2 = steps;
    if (LORA_L3_ABLATE & 32) p1 = steps;
    const int t1 = min((p1 + 2) / 3, turns), t2 = min(max(p2 / 3, t1), turns);
    auto turn = [&](const int p, auto edge_tag) {
        step(p, std::integral_constant<int, 0>{}, edge_tag, std::integral_constant<int, K>{});
        step(p + 1, std::integral_constant<int, 1>{}, edge_tag, std::integral_constant<int, K>{});
        step(p + 2, std::integral_constant<int, 2>{}, edge_tag, std::integral_constant<int, K>{});
    };
    int t = 0;
    if constexpr (!(LORA_L3_ABLATE & 128)) {
        // the first turns of a segment with only the levels that have work (EDGE copies: forcing cells that need none
        // changes nothing); a segment has at least 2 K + 2 steps
        auto fill = [&](auto p_tag) {
            constexpr int p = decltype(p_tag)::value;
            step(p, std::integral_constant<int, p % 3>{}, std::true_type{}, std::integral_constant<int, (p / 2 + 1 < K ? p / 2 + 1 : K)>{});
        };
        fill(std::integral_constant<int, 0>{});
        fill(std::integral_constant<int, 1>{});
        fill(std::integral_constant<int, 2>{});
        t = 1;
        if constexpr (K > 2) {
            fill(std::integral_constant<int, 3>{});
            fill(std::integral_constant<int, 4>{});
            fill(std::integral_constant<int, 5>{});
            t = 2;
        }
    }
    for (; t < t1; ++t) turn(3 * t, std::true_type{});
    for (; t < t2; ++t) turn(3 * t, std::false_type{});
    for (; t < turns; ++t) turn(3 * t, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LORA_L3_STAMP
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        a.stamps[4 * blockIdx.x + 2] = (long long) tx | ((long long) ty << 16) | ((long long) xy_rim << 31);
    }
#endif
    }  // segments
}

template <int TAPSET, int K, int NW>
hipError_t launch_lanes_t(const Plan &p, const double *in, double *out, int begin, int end, int begin2, int end2, hipStream_t s) {
    constexpr int OH = 4 * NW - 2 * K;
    auto kernel = stencil3d_lanes_kernel<TAPSET, K, NW>;
    static int per_cu[64] = {0};  // resolved once per device (and with it the kernel itself: lora_plan_create's share)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (per_cu[dev] == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NW * 64, 0) != hipSuccess || nb < 1) {
            (void) hipGetLastError();
            nb = -1;  // no workgroup of this kernel is resident on this device: the plan must not choose it
        }
        per_cu[dev] = nb;
    }
    if (per_cu[dev] < 0) return hipErrorLaunchOutOfResources;
    if (end <= begin && end2 <= begin2) return hipSuccess;  // prepare_3d_lanes()
    ArgsL3 a{};
    a.in = in;
    a.out = out;
    a.h = p.dims[0];
    a.m = p.dims[1];
    a.n = p.dims[2];
    a.ld = a.n + 8;
    a.plane = (long) (a.m + 4) * (a.n + 8);
    if (a.plane * 8 >= (1L << 31)) return hipErrorInvalidValue;  // 32-bit byte offsets inside a plane
    a.z_begin = begin;
    a.z_end = end;
    a.tiles_x = (a.n + kOutW - 1) / kOutW;
    a.tiles_y = (a.m + OH - 1) / OH;
    const long tiles = (long) a.tiles_x * a.tiles_y;
    // How the launch is cut along z (spans.h): chunks of option fused_z_chunk; else spans (option spans3 = 1, or by itself
    // where they pay: spans_pay); else equal chunks by the model of rounds of workgroups.
    const long slots = (long) std::max(per_cu[dev], 1) * cus, depth = end - begin, depth2 = std::max(end2 - begin2, 0);
    constexpr int S = 2 * K + 1;
    long nblocks = 0;
    a.zc = 0;
    a.z_begin2 = begin2;
    a.z_end2 = begin2 + (int) depth2;
    if (p.fused_z_chunk > 0) {
        a.zc = std::min((long) p.fused_z_chunk, std::max(depth, depth2));
    } else if (depth2 > 0) {  // two ranges (a slab's end regions): chunks, as many workgroups as two ranges of the longer depth
        a.zc = chunk_model(2 * tiles, std::max(depth, depth2), S, slots, 8 * K, nullptr);
    } else {
        const int zc_model = chunk_model(tiles, depth, S, slots, 8 * K, nullptr);
        const bool spans = p.spans3 >= 1 || (p.spans3 < 0 && spans_pay(tiles, depth, S, slots, zc_model, 8.0 * (double) a.plane * (double) depth, 300.0e6));
        if (p.spans3 == 2 && a.tiles_x <= slots) {  // (by option only: measured 4 - 9 % slower than chunks on fp64 grids, spans.h)
            // TEAM spans: the line runs over tile rows and is cut into one piece per team of tiles_x workgroups, one per
            // tile of the row -- x-neighbours stay at the same depth (and, dealt out contiguously, in the same XCD's L2)
            const long nteams = spans_setup(a.sp, 1, a.tiles_y, depth, S, slots / a.tiles_x, 1, 1);
            a.team = a.tiles_x;
            nblocks = nteams * a.tiles_x;
        } else if (spans) {
            nblocks = spans_setup(a.sp, a.tiles_x, a.tiles_y, depth, S, slots, 10, 9);
        }
        if (nblocks == 0) a.zc = zc_model;  // (or a line that does not fit 31 bits of cost units)
    }
    if (a.zc > 0) nblocks = tiles * ((depth + a.zc - 1) / a.zc + (depth2 + a.zc - 1) / a.zc);
    if (a.zc > 0 && depth2 == 0 && p.fused_z_chunk <= 0 && a.tiles_x >= 2 && a.tiles_y >= 2 && !(LORA_L3_ABLATE & 256)) {
        // one round with slots to spare: one chunk more for the tiles of the first tile column and row (see the kernel)
        const long c = (depth + a.zc - 1) / a.zc, nslow = a.tiles_x + a.tiles_y - 1;
        const long slow_zc = (depth + c) / (c + 1);
        if (nblocks <= slots && nblocks + nslow <= slots && slow_zc >= 2 * K) {
            a.slow_c = (int) ((depth + slow_zc - 1) / slow_zc);
            a.slow_zc = (int) slow_zc;
            nblocks = nslow * a.slow_c + (tiles - nslow) * c;
        }
    }
    if (a.zc > 0 && a.slow_c == 0 && nblocks > slots && !(LORA_L3_ABLATE & 512)) {  // several rounds: deal the chunks out
        if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
        a.deal = (int) nblocks;
        nblocks = slots;
    }
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
#ifdef LORA_L3_STAMP
    extern long long *g_l3_stamps;
    extern long g_l3_stamp_blocks;
    a.stamps = g_l3_stamps;
    g_l3_stamp_blocks = nblocks;
#endif
    Taps27 w;
    for (int k = 0; k < 27; ++k) w.w[k] = p.w[k];
    if (TAPSET == TAPS3D_SEP)
        for (int k = 0; k < 9; ++k) w.w[k] = p.sep64[k];
    hipLaunchKernelGGL(kernel, dim3((unsigned) nblocks), dim3(NW * 64), 0, s, a, w);
    return hipGetLastError();
}

}  // namespace

// K = 4 (or 2) applications in one launch over interior planes [begin, end); star taps or exactly separable box taps
// (p.sep64_valid), reference boundary
// ... and, in the same launch, planes [begin2, end2) (empty by default; a slab's two end regions cost one pipeline start
// each either way, but one launch of 13 steps instead of two)
hipError_t launch_3d_lanes(const Plan &p, int K, const double *in, double *out, int begin, int end, hipStream_t s, int begin2, int end2) {
    if (p.boundary != LORA_BC_REFERENCE || p.dtype != LORA_F64) return hipErrorNotSupported;
    const bool sep = p.tapset == TAPS3D_BOX && p.sep64_valid;
    if (p.tapset != TAPS3D_STAR && !sep) return hipErrorNotSupported;
    if (K == 4)
        return sep ? launch_lanes_t<TAPS3D_SEP, 4, 8>(p, in, out, begin, end, begin2, end2, s)
                   : launch_lanes_t<TAPS3D_STAR, 4, 8>(p, in, out, begin, end, begin2, end2, s);
    if (K == 2)
        return sep ? launch_lanes_t<TAPS3D_SEP, 2, 8>(p, in, out, begin, end, begin2, end2, s)
                   : launch_lanes_t<TAPS3D_STAR, 2, 8>(p, in, out, begin, end, begin2, end2, s);
    return hipErrorInvalidValue;
}

// one-time host work (kernel resolution, residency query) of the plan's instantiations; no launch.  false: a device is
// there and says no workgroup of the kernel fits it -- the plan then keeps the tile kernels (capi.cpp: plan_refresh).
bool prepare_3d_lanes(const Plan &p) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void) hipGetLastError();
        return true;  // no device here (a build box): plans resolve as they would on the GPU
    }
    const hipError_t e4 = launch_3d_lanes(p, 4, nullptr, nullptr, 0, 0, nullptr);
    const hipError_t e2 = launch_3d_lanes(p, 2, nullptr, nullptr, 0, 0, nullptr);
    return e4 == hipSuccess && e2 == hipSuccess;
}

const char *kernel_name_3d_lanes(const Plan &) { return "stencil3d_lanes_kernel"; }

}  // namespace lora
