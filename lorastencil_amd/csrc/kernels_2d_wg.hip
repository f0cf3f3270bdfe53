// kernels_2d_wg.hip -- K = KL x S applications of a radius-3 2D stencil per launch: workgroup-wide rows, the K time
// levels pipelined over S groups of four waves ("stages").
//
// Where round 2 left the headline (counters profiles/r02_star2d1r_final2_pmc.json): the row-streaming
// kernel of kernels_2d_stream.hip runs four levels inside every wave on a PRIVATE strip of 128 input columns.  Its
// instruction stream is 88 % fp64 arithmetic, but
//   * 24 of a strip's 128 columns (19 %) are halo that every level recomputes and the launch discards;
//   * four levels of seven rotating partial rows are 112 of its 130 VGPRs: three waves per SIMD, and the occupancy
//     counter reads 2.47 because 5.9 workgroups per CU run as two ragged rounds;
//   * a wave that both loads and stores cannot tell its own loads from its stores in vmcnt, so "row r has landed" must be
//     waited for as "all but D - 1 vector-memory operations done" -- which also waits for the next row;
//   * at four sweeps per pass the launch still moves 4.46 GB in 1.24 ms = 3.6 TB/s beside a 65 % busy vector pipe.
// The matrix pipe is no way out for fp64: v_mfma_f64_16x16x4 and v_fma_f64 share one datapath on gfx950 -- issued
// together, from one wave or from two waves of a SIMD, their times ADD (tools/probes/fp64_coissue_probe.hip,
// profiles/r03_fp64_coissue_probe.txt: 64 + 81 -> 149 cycles).
//
// This kernel keeps the scatter evaluation of rows_2d.h and the ring of LDS-DMA'd input rows, and changes the ownership:
//   * A ROW belongs to the WORKGROUP: 512 input columns = one 16-byte piece per lane of four waves.  Every level shifts
//     the lane -> column map by 3 (aligned 8-wide LDS windows, as before), but now once per workgroup: a strip yields
//     512 - 6 K output columns (476 of 512 = 93 % at K = 6; 104 of 128 = 81 % before).  A lane's window reaches 6 columns
//     into its right neighbour's piece -- also the next wave's -- so every level row goes through a workgroup-shared LDS
//     row buffer, two copies per level (written in step r, read in step r + 1), and ONE s_barrier per step orders it all.
//   * The K levels are split over S stages of KL levels: waves 4 s .. 4 s + 3 run levels s KL + 1 .. (s + 1) KL of the same
//     512 columns.  The hand-off between stages is the same row buffer every level already goes through, so a stage
//     costs nothing extra in LDS traffic, and a wave carries KL x 7 partial rows instead of K x 7: K = 6 in 2 x 3 levels
//     fits 128 VGPRs = four waves per SIMD (two 8-wave workgroups per CU) where six levels in one wave would need 190.
//   * Stage 0 issues the loads (global -> LDS, one 1 KiB piece per wave and row, D rows ahead), the last stage the stores:
//     no wave has both in its vmcnt, so "row r has landed" is exactly vmcnt(D - 1) and stores are never waited for.
//   * The launch is ONE round: strips x chunks = the workgroups that are resident at once (2 per CU), every chunk as
//     long as that allows (16384^2: 35 strips x 14 chunks of 1171 rows, 7 K - 1 = 41 recomputed steps each = 3.5 %).
//   * Halo semantics (as kernels_2d_stream.hip): cells of an intermediate level outside the interior are 0 at
//     odd levels and the source buffer's own halo value at even levels.  Only workgroups at the rim of the grid can see
//     such cells: they run a second copy of the step (EDGE) that forces them.  The halo values come straight from the
//     input array -- while fused launches run, every buffer carries buffer 0's halo (capi.cpp) -- but NOT through a load
//     the consuming wave waits for: vmcnt completes in order, so that wait would also drain the rows in flight (stage 0) or
//     the store just issued (last stage), every step (measured: rim workgroups 2.3 x slower).  Instead stage 0 fetches
//     them one step ahead, global -> LDS like the rows, into a row of halo values per even level (range-checked offsets:
//     lanes that need nothing make no memory request), where every stage picks them up behind the step's barrier.
//     Under the Dirichlet option every level keeps the source's values: a row of halo values per level (K = 4 and 2).
// Per accumulator the taps arrive in the same order as in the other fused 2D kernels: results are bit-identical to them.
//
// Replaces the reference's time-step loop 2d/gpu.cu:544-546 (K steps per pass) and kernels 2d/gpu.cu:31-273.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_common.h"
#include "rows_2d.h"

namespace lora {

struct ArgsWG {
    const double *in;
    double *out;
    int ld, m, n;
    int row_begin, row_end;
    int strips;  // column strips of outw output columns
    int outw;    // output columns per strip: 512 - 6 K
    // Row chunks.  The first and the last strip ("rim strips": their lanes see columns outside the interior in every step)
    // run the slower EDGE loop throughout and get shorter chunks, so that every workgroup of the one round takes about
    // the same time: rim strips have chunks_e chunks of rows_e rows, the others chunks_i of rows_i.
    int rim;  // number of rim strips in this launch: 2, or all of them when there are fewer than three
    int rows_i, groups_i, chunks_i;
    int rows_e, groups_e, chunks_e;
    int prio_split;  // 0, or the number of workgroups dispatched first (one per CU): see the time-sliced priorities in the kernel
    int prio_shift;  // a priority slice is 2^prio_shift ticks of the 100 MHz counter
#ifdef LORA_DIAGNOSTICS
    long long *stamps;  // per workgroup: s_memrealtime at entry / exit, HW_ID, XCC_ID (tools/probes/wg_stamps.hip only)
#endif
};

namespace {

constexpr int kRowW = 512;  // doubles of a workgroup row: 4 waves x 64 lanes x 2 columns
constexpr int kBufW = 520;  // doubles of a level row buffer: the last lanes' windows run 6 doubles past the row

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ring of input rows | two copies of a row per intermediate level | two copies of a row of halo values per even level
// (DIRI: the Dirichlet option -- every intermediate level keeps the source's halo values, not only the even ones)
__host__ __device__ constexpr int wg_halo_levels(int K, bool DIRI) { return DIRI ? K - 1 : K / 2 - 1; }
__host__ __device__ constexpr int wg_lds_doubles(int K, int D, bool DIRI) { return (D + 1) * kRowW + (K - 1) * 2 * kBufW + wg_halo_levels(K, DIRI) * 2 * kRowW + 8; }

// Step r of a chunk whose first output row is i0: stage 0 consumes input row i0 - 3 K + r; level l completes its row
// i0 - 3 K - 4 l + 1 + r (a level lags 3 rows -- its radius -- plus one step of hand-off behind the level below).
// Level K is the output.  Ring slot of input row r: r mod (D + 1); row buffer of level l written in step r: copy r mod 2.
template <int EVAL, int KL, int S, int D, int WPS, bool DIRI>
__global__ __launch_bounds__(256 * S, WPS) void stencil2d_wg_kernel(const ArgsWG a, const Taps49 W, const LowRankTaps F) {
    constexpr int K = KL * S, NS = D + 1;
    static_assert(NS == 4 || NS == 2, "ring slots: a power of two");
    static_assert(K % 2 == 0, "fused launches move an even number of levels");
    __shared__ __attribute__((aligned(128))) double lds[wg_lds_doubles(K, D, DIRI)];
    double *const ring = lds;
    double *const rb = lds + NS * kRowW;  // rb + ((l - 1) * 2 + copy) * kBufW: row of level l

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stage = S == 1 ? 0 : (wv >> 2);
    const int wq = wv & 3;
    const int t = wq * 64 + lane;  // this lane's column pair of the workgroup's rows
    const bool first = stage == 0;  // stage s runs levels s KL + 1 .. (s + 1) KL

    // workgroup -> (strip, chunk): rim strips first
    const int blk = xcd_contiguous(blockIdx.x, gridDim.x);
    int strip, chunk, rows, groups;
    if (blk < a.rim * a.chunks_e) {
        const int q = blk / a.chunks_e;
        chunk = blk - q * a.chunks_e;
        strip = a.rim == 2 ? (q == 0 ? 0 : a.strips - 1) : q;
        rows = a.rows_e;
        groups = a.groups_e;
    } else {
        const int b2 = blk - a.rim * a.chunks_e, ni = a.strips - a.rim;
        chunk = b2 / ni;
        strip = 1 + (b2 - chunk * ni);
        rows = a.rows_i;
        groups = a.groups_i;
    }
#ifdef LORA_DIAGNOSTICS
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        a.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(63492) | ((long long) __builtin_amdgcn_s_getreg(63508) << 32);
        a.stamps[4 * blockIdx.x + 3] = ((long long) strip << 32) | (unsigned) chunk;
    }
#endif
    const int i0 = a.row_begin + chunk * rows;  // first output row of the chunk (interior coordinates)
    const int j0 = strip * a.outw;              // first output column of the strip
    const int row_hi = min(i0 + rows, a.row_end);
    groups = min(groups, (row_hi - i0 + 7 * K - 1 + 6) / 7);  // the last chunk of a strip may be shorter
    // Can this workgroup's lanes see a column outside the interior at some level?  (rim strips; uniform)
    const bool col_edge = j0 - 3 * K < 0 || j0 + kRowW > a.n;

    // input: interior columns j0 - 3 K + 2 t, +1 = padded columns j0 - 3 K + 4 + 2 t, clamped into the padded array
    // (clamped pieces only feed cells outside the interior, which the EDGE steps force)
    const int gcol = min(max(j0 - 3 * K + 4 + 2 * t, 0), a.n + 6);
    auto issue = [&](int r) {
        const int pr = min(max(i0 - 3 * K + 4 + r, 0), a.m + 7);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (a.in + (size_t) pr * a.ld + gcol),
                                         (__attribute__((address_space(3))) void *) (ring + (r & (NS - 1)) * kRowW + wq * 128), 16, 0, 0);
    };
    // stores: lanes t < outw / 2 write interior columns j0 + 2 t, +1; the descriptor's range check drops the rest
    const unsigned store_off = t < a.outw / 2 ? 16u * t : 0x80000000u;
    const unsigned row_bytes = (unsigned) (max(min(a.outw, a.n - j0), 0) * 8);
    double *const out_strip = a.out + (j0 + 4);

    // the row buffers start as zeros (the first 6 rows of every level are partial sums that nothing uses)
    for (int k = threadIdx.x; k < (K - 1) * 2 * kBufW / 2 + 4; k += 256 * S) *reinterpret_cast<d2 *>(rb + 2 * k) = (d2){0.0, 0.0};
    if (first)
#pragma unroll
        for (int k = 0; k < D; ++k) issue(k);

    // One copy of the 7-step group per (stage, EDGE): inside it the level numbers, the LDS addresses and the roles (who
    // loads, who stores) are compile-time constants and the instruction stream has NO branches -- with branches in it the
    // compiler sinks the partial-sum updates of a row towards their use, 7 steps later, and keeps every window alive
    // meanwhile (256 VGPRs and scratch instead of 128).  EDGE is chosen per group and wave: rim strips always, the other
    // strips only in the few groups whose level rows lie outside the interior (first / last chunk); both copies have
    // one barrier per step, so waves of one workgroup may run different copies.
    auto run_stage = [&](auto stage_tag) {
        constexpr int STAGE = decltype(stage_tag)::value;
        constexpr bool FIRST = STAGE == 0;
        constexpr int LB = STAGE * KL;                             // levels LB + 1 .. LB + KL
        double pa[KL][7], pb[KL][7];  // rotating partial sums of this wave's KL levels, columns 2 t / 2 t + 1
#pragma unroll
        for (int j = 0; j < KL; ++j)
#pragma unroll
            for (int k = 0; k < 7; ++k) pa[j][k] = pb[j][k] = 0.0;
        const double *const winp = lds + 2 * t;  // + constant offsets: every window and row-buffer access of this lane
        // EDGE: per intermediate level of this wave, which of the lane's two columns are interior columns
        // (as all-ones / zero masks)
        unsigned cmask0[KL], cmask1[KL];
#pragma unroll
        for (int j = 0; j < KL; ++j) {
            const int c0 = j0 - 3 * K + 3 * (LB + j + 1) + 2 * t;
            cmask0[j] = (unsigned) c0 < (unsigned) a.n ? ~0u : 0u;
            cmask1[j] = (unsigned) (c0 + 1) < (unsigned) a.n ? ~0u : 0u;
            asm volatile("" : "+v"(cmask0[j]), "+v"(cmask1[j]));
        }
        // Stage 0, EDGE: per even level 2 e + 2 (of ANY stage) the byte offset of this lane's cell in a padded row when the
        // lane needs the source's value there (0x80000000 = beyond the descriptor's range: no memory request) --
        // `hcol` in rows of the interior (rim columns only), `hall` in rows outside it (every lane)
        constexpr int NH = wg_halo_levels(K, DIRI);  // levels with halo values: level DIRI ? e + 1 : 2 e + 2 is number e
        unsigned hcol[NH > 0 ? NH : 1], hall[NH > 0 ? NH : 1];
        if (FIRST) {
#pragma unroll
            for (int e = 0; e < NH; ++e) {
                const int c0 = j0 - 3 * K + 3 * (DIRI ? e + 1 : 2 * e + 2) + 2 * t;
                hall[e] = 8u * (unsigned) min(max(c0 + 4, 0), a.n + 6);
                hcol[e] = ((unsigned) c0 < (unsigned) a.n && (unsigned) (c0 + 1) < (unsigned) a.n) ? 0x80000000u : hall[e];
            }
        }
        double *const hb = lds + NS * kRowW + (K - 1) * 2 * kBufW;  // hb + (e * 2 + copy) * kRowW: halo values of level 2 e + 2

        auto step = [&](const int r, auto phase_tag, auto edge_tag) {
            constexpr int P = decltype(phase_tag)::value;  // r mod 7: logical row 0 (the one completed now) is acc[P]
            constexpr bool EDGE = decltype(edge_tag)::value;
            const int par = r & 1;
            // "input row r has landed": the D - 1 younger row loads may be outstanding (S > 1: stage 0 has no stores in its
            // vmcnt; S == 1: as in kernels_2d_stream.hip, the interleaved stores only make the wait stricter).  EDGE: this
            // step's halo values were issued in the previous step, BEFORE its row load: one younger load.
            if (FIRST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EDGE && NH > 0 ? 1 : D - 1) : "memory");
            // one barrier per step: behind it every wave's piece of input row r (and of this step's halo values) and every
            // level row of step r - 1 is visible, and nobody still reads what this step overwrites (ring slot of row r - 1,
            // row copies of step r - 2, halo values of step r - 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int cur = par * kBufW, prev = kBufW - cur;  // this step's / the previous step's copy of a level row
#pragma unroll
            for (int j = KL - 1; j >= 0; --j) {
                const int l = LB + j + 1;                    // global level (a constant after unrolling)
                const int row = i0 - 3 * K - 4 * l + 1 + r;  // the row this level completes now
                const bool from_ring = l == 1;               // level 1 reads the input row
                const bool to_global = l == K;               // level K is the output
                const double *const src = from_ring ? winp + (r & (NS - 1)) * kRowW : winp + NS * kRowW + (l - 2) * 2 * kBufW + prev;
                double win[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const d2 w2 = *reinterpret_cast<const d2 *>(src + 2 * q);
                    win[2 * q] = w2.x;
                    win[2 * q + 1] = w2.y;
                }
                // EDGE: what this lane's cells are forced to when they lie outside the interior
                d2 h = {0.0, 0.0};
                if (EDGE && !to_global && ((l & 1) == 0 || DIRI))
                    h = *reinterpret_cast<const d2 *>(winp + NS * kRowW + (K - 1) * 2 * kBufW + ((DIRI ? l - 1 : l / 2 - 1) * 2 + par) * kRowW);
                // the newest logical row (6) starts from zero: stated here, so that the zero is an inline constant of its
                // first multiply-add and not a register carried around the loop
                pa[j][(P + 6) % 7] = 0.0;
                pb[j][(P + 6) % 7] = 0.0;
                apply_row<EVAL, 7, P>(6, win, pa[j], pb[j], W, F);
                d2 v;
                v.x = pa[j][P];
                v.y = pb[j][P];
                if (to_global) {
                    const bool live = row >= i0 && row < row_hi;
                    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
                        out_strip + (size_t) (max(row, 0) + 4) * a.ld, 0, live ? row_bytes : 0u, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst, store_off, 0, kStoreNT);
                } else {
                    if (EDGE) {
                        // v = inside ? v : h as a bitwise blend under integer masks the compiler cannot see through: from a
                        // select it builds a branch around the row's last multiply-add, and with branches in the step the
                        // register blow-up is back
                        const unsigned rmask = (unsigned) row < (unsigned) a.m ? ~0u : 0u;
                        const unsigned m0 = cmask0[j] & rmask, m1 = cmask1[j] & rmask;
                        const u32x4 vb = __builtin_bit_cast(u32x4, v), hb4 = __builtin_bit_cast(u32x4, h);
                        u32x4 o;
                        o.x = (vb.x & m0) | (hb4.x & ~m0);
                        o.y = (vb.y & m0) | (hb4.y & ~m0);
                        o.z = (vb.z & m1) | (hb4.z & ~m1);
                        o.w = (vb.w & m1) | (hb4.w & ~m1);
                        v = __builtin_bit_cast(d2, o);
                    }
                    *reinterpret_cast<d2 *>(const_cast<double *>(winp) + NS * kRowW + (l - 1) * 2 * kBufW + cur) = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // The compiler does not know that the LDS-DMAs below overwrite LDS; nothing of this step may sink below them
            asm volatile("" ::: "memory");
            if (FIRST) {
                if (EDGE) {
                    // halo values of step r + 1, every even level: this wave's 128 columns of them
#pragma unroll
                    for (int e = 0; e < NH; ++e) {
                        const int row1 = i0 - 3 * K - 4 * (DIRI ? e + 1 : 2 * e + 2) + 1 + (r + 1);
                        const int pr = min(max(row1 + 4, 0), a.m + 7);
                        const __amdgpu_buffer_rsrc_t hsrc = __builtin_amdgcn_make_buffer_rsrc(
                            const_cast<double *>(a.in) + (size_t) pr * a.ld, 0, (unsigned) a.ld * 8u, 0x00020000);
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass does not know this builtin and then drops the kernel's stub without a word)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            hsrc, (__attribute__((address_space(3))) void *) (hb + (e * 2 + (par ^ 1)) * kRowW + wq * 128), 16,
                            (unsigned) row1 < (unsigned) a.m ? hcol[e] : hall[e], 0, 0, 0);
#else
                        (void) hsrc;
                        (void) hb;
#endif
                    }
                }
                issue(r + D);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto group = [&](const int r0, auto edge_tag) {
            step(r0 + 0, std::integral_constant<int, 0>{}, edge_tag);
            step(r0 + 1, std::integral_constant<int, 1>{}, edge_tag);
            step(r0 + 2, std::integral_constant<int, 2>{}, edge_tag);
            step(r0 + 3, std::integral_constant<int, 3>{}, edge_tag);
            step(r0 + 4, std::integral_constant<int, 4>{}, edge_tag);
            step(r0 + 5, std::integral_constant<int, 5>{}, edge_tag);
            step(r0 + 6, std::integral_constant<int, 6>{}, edge_tag);
        };

        // Rows the workgroup's intermediate levels complete in group g: [lo0 + 7 g, hi0 + 7 g].  Groups [0, g1) and
        // [g2, groups) have rows outside the interior (first / last chunk only); rim strips run EDGE throughout.  The second
        // EDGE range starts one group early: its first step finds no halo values fetched for it, and needs none there.
        // The ranges are the same for every wave of the workgroup (stage 0 fetches for all).  Three plain loops: a branch
        // inside one loop brings the register blow-up described above back.
        const int lo0 = i0 - 3 * K - 4 * (K - 1) + 1, hi0 = i0 - 3 * K - 4 + 1 + 6;
        int g1 = lo0 >= 0 ? 0 : (-lo0 + 6) / 7, g2 = hi0 >= a.m ? 0 : (a.m - hi0 + 6) / 7 - 1;
        if (col_edge) g1 = groups;
        g1 = min(g1, groups);
        g2 = min(max(g2, g1), groups);
        // Two workgroups share a CU, and the SIMD arbiter favours the OLDER wave: left alone, the first-dispatched workgroup
        // runs at nearly full speed, finishes at 60 % of the launch, and the other one runs the rest of the time without a
        // partner (measured timeline: tools/probes/wg_stamps.hip -- 730 us and 1180 us on every CU; a workgroup alone
        // takes 810).  Time-sliced priorities share the CU evenly instead: every group of 7 steps a wave sets its
        // priority to one bit of the 100 MHz real-time counter (it flips every 2^prio_shift ticks), inverted in the
        // workgroups dispatched second (block index >= prio_split = number of CUs), so that the two workgroups of a CU
        // hold complementary priorities and swap them every slice: both progress at the same average rate, finish
        // together, and neither runs without a partner.
        const int cls = a.prio_split > 0 ? ((int) blockIdx.x >= a.prio_split ? 1 : 0) : -1;
        auto slice = [&]() {
            if (cls >= 0) {
                const int bit = (int) (__builtin_amdgcn_s_memrealtime() >> a.prio_shift) & 1;
                if (bit != cls)
                    __builtin_amdgcn_s_setprio(1);
                else
                    __builtin_amdgcn_s_setprio(0);
            }
        };
        int g = 0;
        if constexpr (STAGE > 0) {
            // Level l needs its first row complete in step 7 l - 1 and starts accumulating it six steps earlier: the levels
            // of stage s have nothing to do in the first s KL groups of a chunk.  Their waves only keep the barrier count
            // there, and the stages below have the vector pipe to themselves (a chunk costs about rows + 30 steps instead
            // of rows + 41 at K = 6: what thin shares of a multi-GPU run and small grids lose most).
            const int idle = min(STAGE * KL, groups);
            for (int q = 0; q < 7 * idle; ++q) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            g = idle;
        }
        for (; g < g1; ++g) {
            slice();
            group(7 * g, std::true_type{});
        }
        for (; g < g2; ++g) {
            slice();
            group(7 * g, std::false_type{});
        }
        for (; g < groups; ++g) {
            slice();
            group(7 * g, std::true_type{});
        }
    };
    static_assert(S >= 1 && S <= 4, "stages");
    if (stage == 0) run_stage(std::integral_constant<int, 0>{});
    if constexpr (S >= 2)
        if (stage == 1) run_stage(std::integral_constant<int, 1>{});
    if constexpr (S >= 3)
        if (stage == 2) run_stage(std::integral_constant<int, 2>{});
    if constexpr (S >= 4)
        if (stage == 3) run_stage(std::integral_constant<int, 3>{});
#ifdef LORA_DIAGNOSTICS
    if (a.stamps && threadIdx.x == 0) a.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    // drain: the last D loads target this workgroup's LDS, which the next workgroup on this CU may own
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Workgroups of this instantiation that are resident per CU (cached per device).  The first query of an instantiation
// also makes the runtime load and resolve the kernel -- half a millisecond of host time that lora_plan_create /
// lora_plan_set_* pay through prepare_2d_wg(), not the first launch of a run.
template <int EVAL, int KL, int S, int D, int WPS, bool DIRI>
int wg_per_cu(int dev) {
    static int per_cu[64] = {0};
    if (dev < 0 || dev >= 64) dev = 0;
    if (per_cu[dev] == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stencil2d_wg_kernel<EVAL, KL, S, D, WPS, DIRI>, 256 * S, 0) != hipSuccess || nb < 1) {
            (void) hipGetLastError();
            return 1;  // (no device: asked again later)
        }
        per_cu[dev] = nb;
    }
    return per_cu[dev];
}

template <int EVAL, int KL, int S, int D, int WPS, bool DIRI>
hipError_t launch_wg_t(const Plan &p, ArgsWG a, int rows_total, hipStream_t s) {
    constexpr int K = KL * S;
    auto kernel = stencil2d_wg_kernel<EVAL, KL, S, D, WPS, DIRI>;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    // one round: as many workgroups as are resident at once
    const int per_cu_dev = wg_per_cu<EVAL, KL, S, D, WPS, DIRI>(dev);
    if (rows_total <= 0) return hipSuccess;  // prepare_2d_wg(): the query above is all that was wanted
    const int lag = 7 * K - 1;  // steps of a chunk beyond its output rows
    auto fit = [&](long rows) {
        long g = (rows + lag + 6) / 7;
        if (g < K + 1) g = K + 1;
        return (int) (7 * g - lag);
    };
    a.rim = a.strips >= 3 ? 2 : a.strips;
    const int ni = a.strips - a.rim;
    // A rim-strip workgroup runs the EDGE steps throughout (selects, halo-value fetches, one row less in flight) and takes
    // about half as long again per step: its chunks are that much shorter (16384^2, us per launch at 12 / 25 / 35 / 45 /
    // 55 / 70 per cent: 1393 / 1391 / 1237 / 1189 / 1172 / 1164; gpurun_out/wg_sweep.log)
    const int pct = p.wg_edge_pct >= 0 ? p.wg_edge_pct : 60;
    auto rim_rows = [&](int rows_i) { return fit(std::max(1L, (long) (rows_i + lag) * 100 / (100 + pct) - lag)); };
    if (p.wg_rows > 0) {
        a.rows_i = fit(p.wg_rows);
    } else {
        // ONE round: the most interior chunks for which every workgroup of the launch is resident at once
        const long slots = (long) per_cu_dev * cus;
        // Small grids: filling the round matters more than the rows a chunk recomputes -- down to chunks of about 1.5 x
        // the 7 K - 1 warm-up rows (star2d1r, six sweeps, GStencils/s at chunks of 64 / 100 / 128 / 164 / 256 rows: 4096^2
        // 636 / 837 / 762 / 675 / 595, 2048^2 375 / 289 / 245 / 208 / 149, 1024 x 16384 685 / 875 / 775 / 816 / 595 --
        // tools/wg_small.py; the first rule, four times the warm-up, left half the CUs idle on such grids)
        const long min_rows = 3 * lag / 2;
        int best = fit(rows_total);
        for (long c = 1; c <= rows_total; ++c) {
            const int ri = fit((rows_total + c - 1) / c);
            if (ri < min_rows && c > 1) break;
            const int re = ni > 0 ? rim_rows(ri) : ri;
            const long wgs = (long) ni * ((rows_total + ri - 1) / ri) + (long) a.rim * ((rows_total + re - 1) / re);
            if (wgs > slots && c > 1) break;
            best = ri;
        }
        a.rows_i = best;
    }
    a.rows_e = ni > 0 ? rim_rows(a.rows_i) : a.rows_i;
    a.groups_i = (a.rows_i + lag + 6) / 7;
    a.groups_e = (a.rows_e + lag + 6) / 7;
    a.chunks_i = (rows_total + a.rows_i - 1) / a.rows_i;
    a.chunks_e = (rows_total + a.rows_e - 1) / a.rows_e;
    a.prio_split = (p.wg_prio != 0 && per_cu_dev == 2) ? cus : 0;
    a.prio_shift = p.wg_prio > 0 ? p.wg_prio : 12;
    Taps49 w;
    for (int k = 0; k < 49; ++k) w.w[k] = p.w[k];
    LowRankTaps f{};
    for (int tt = 0; tt < 3; ++tt)
        for (int e = 0; e < 7; ++e) {
            f.u[tt][e] = p.lowrank.u[tt][e];
            f.v[tt][e] = p.lowrank.v[tt][e];
        }
    f.rc = p.lowrank_rc;
    if (EVAL == EVAL_NEST)
        for (int k = 0; k < 4; ++k) {
            f.u[0][k] = p.nest_g[k];
            f.v[0][k] = p.nest_a[k];
        }
    const long nblocks = (long) ni * a.chunks_i + (long) a.rim * a.chunks_e;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kernel, dim3((unsigned) nblocks), dim3(256 * S), 0, s, a, w, f);
    return hipGetLastError();
}

// This file is compiled once per launch depth (LORA_WG_K = 6, 4, 2: eight tap evaluations each).  Two stages of K / 2
// levels, three rows in flight, four waves per SIMD in every case.
#ifndef LORA_WG_K
#define LORA_WG_K 6
#endif
#ifndef LORA_WG_S  // stages: 2 (two 8-wave workgroups per CU) for K <= 6; 4 (one 16-wave workgroup per CU) for K = 12 and 8
#define LORA_WG_S 2
#endif
template <int EVAL>
hipError_t launch_wg_e(const Plan &p, const ArgsWG &a, int rows_total, hipStream_t s) {
#if LORA_WG_K <= 4  // the Dirichlet option: halo values at every level (K = 6 would need 98 KB of LDS per workgroup)
    if (p.boundary == LORA_BC_DIRICHLET) return launch_wg_t<EVAL, LORA_WG_K / 2, 2, 3, 4, true>(p, a, rows_total, s);
#endif
    return launch_wg_t<EVAL, LORA_WG_K / LORA_WG_S, LORA_WG_S, 3, 4, false>(p, a, rows_total, s);
}

}  // namespace

#define LORA_WG_CAT2(a, b) a##b
#define LORA_WG_CAT(a, b) LORA_WG_CAT2(a, b)
#define LORA_WG_ENTRY LORA_WG_CAT(launch_2d_wg_k, LORA_WG_K)

// LORA_WG_K applications in one launch over interior rows [begin, end)
hipError_t LORA_WG_ENTRY(const Plan &p, ArgsWG a, int rows_total, hipStream_t s) {
    switch (p.fused_eval) {
        case EVAL_NEST:
            return launch_wg_e<EVAL_NEST>(p, a, rows_total, s);
        case EVAL_LR_DIAMOND:
            return launch_wg_e<EVAL_LR_DIAMOND>(p, a, rows_total, s);
        case EVAL_LR_PYRAMID:
            return launch_wg_e<EVAL_LR_PYRAMID>(p, a, rows_total, s);
        case EVAL_LR_PYRAMID_SYM:
            return launch_wg_e<EVAL_LR_PYRAMID_SYM>(p, a, rows_total, s);
        case EVAL_LR_PYRAMID_SYM_GAP:
            return launch_wg_e<EVAL_LR_PYRAMID_SYM_GAP>(p, a, rows_total, s);
        default:
            break;
    }
    switch (p.tapset) {
        case TAPS2D_DIAMOND:
            return launch_wg_e<TAPS2D_DIAMOND>(p, a, rows_total, s);
        case TAPS2D_STAR:
            return launch_wg_e<TAPS2D_STAR>(p, a, rows_total, s);
        default:
            return launch_wg_e<TAPS2D_BOX>(p, a, rows_total, s);
    }
}

#if LORA_WG_K == 6
hipError_t launch_2d_wg_k4(const Plan &p, ArgsWG a, int rows_total, hipStream_t s);
hipError_t launch_2d_wg_k2(const Plan &p, ArgsWG a, int rows_total, hipStream_t s);

// One-time host work of the three instantiations this plan's runs launch (six applications and the four / two of the
// tails): kernel resolution + the residency query.  No launch; without a device nothing happens.
void prepare_2d_wg(const Plan &p) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void) hipGetLastError();
        return;
    }
    ArgsWG a{};
    (void) launch_2d_wg_k6(p, a, 0, nullptr);
#ifndef LORA_WG_ONLY_K6
    (void) launch_2d_wg_k4(p, a, 0, nullptr);
    (void) launch_2d_wg_k2(p, a, 0, nullptr);
#endif
}

int wg_strip_width(int K) { return kRowW - 6 * K; }

#ifdef LORA_DIAGNOSTICS
long long *g_wg_stamps = nullptr;  // set by the probe: 4 x int64 per workgroup
#endif

// K = 6, 4 or 2 applications in one launch over interior rows [begin, end)
hipError_t launch_2d_wg(const Plan &p, int K, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    if (K != 6 && K != 4 && K != 2) return hipErrorInvalidValue;
    if (p.boundary == LORA_BC_PERIODIC || (p.boundary == LORA_BC_DIRICHLET && K == 6)) return hipErrorNotSupported;
    ArgsWG a;
    a.in = in;
    a.out = out;
    a.m = p.dims[0];
    a.n = p.dims[1];
    a.ld = a.n + 8;
    a.row_begin = begin;
    a.row_end = end;
    a.outw = wg_strip_width(K);
    a.strips = (a.n + a.outw - 1) / a.outw;
    a.rim = a.rows_i = a.groups_i = a.chunks_i = a.rows_e = a.groups_e = a.chunks_e = 0;
    a.prio_split = a.prio_shift = 0;
#ifdef LORA_DIAGNOSTICS
    a.stamps = g_wg_stamps;
#endif
#ifdef LORA_WG_ONLY_K6  // (tools/probes/wg_stamps.hip includes this file alone)
    return launch_2d_wg_k6(p, a, end - begin, s);
#else
    return K == 6 ? launch_2d_wg_k6(p, a, end - begin, s)
                  : (K == 4 ? launch_2d_wg_k4(p, a, end - begin, s) : launch_2d_wg_k2(p, a, end - begin, s));
#endif
}

const char *kernel_name_2d_wg(const Plan &) { return "stencil2d_wg_kernel"; }
#endif  // LORA_WG_K == 6

}  // namespace lora
