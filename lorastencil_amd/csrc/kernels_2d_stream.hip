// kernels_2d_stream.hip -- FOUR (or two) kernel applications of a radius-3 2D stencil per launch, row-streaming form.
// (Round 3: the default for 2D is now kernels_2d_wg.hip -- six applications, workgroup-wide rows; this kernel serves
// plans that ask for four or two applications per launch, the slab drivers, and the Dirichlet option.)
//
// The tile kernel of kernels_2d_fused.hip holds a whole 46 x 136 input window in LDS per workgroup: 50 KB, three
// workgroups per CU, and inside a workgroup strictly serial phases (window load -> barrier -> application 1 ->
// barrier -> application 2 + stores).  It moves only the compulsory HBM bytes but at 4.6-5.0 TB/s (PMC, profiles/),
// below the 5.6 TB/s the one-sweep kernel streams at: while a workgroup computes it has no loads in flight, 18 % of
// its level-1 rows and 35 % of its window rows are recomputed / re-read halo, and a third of its vector instructions
// are tile bookkeeping (clamped addresses, guards).
//
// Here every WAVE is autonomous and walks DOWN a column strip (the time-step loop of the reference driver,
// 2d/gpu.cu:544-546, K = 2 or 4 steps per pass; kernel replaced: 2d/gpu.cu:181-273):
//   * a strip is 128 input columns = 128 - 3 l columns of level l = 128 - 6 K output columns (116 for K = 2, 104 for
//     K = 4): exactly one 16-byte piece per lane and row, so a row of the strip is ONE global_load_lds_dwordx4 (1 KiB,
//     global -> LDS, no staging registers) and one buffer_store_dwordx4;
//   * the scatter form of apply_row needs each input row once: LDS only serves as the cross-lane window exchange of
//     the row being consumed (4 x ds_read_b128 per row and level, as in the tile kernel), so a wave owns a ring of
//     seven (K = 2) or ten (K = 4) 1 KiB input-row slots (rows in flight, and the halo source rows of level 2) plus one
//     row buffer per intermediate level: 8.3 / 13.3 KB per wave, 16 / 12 waves per CU;
//   * the seven partially summed rows of each level live in rotating register files (the loop is unrolled by 7 so
//     that the rotation is a compile-time renaming, apply_row<.., ROT>): 76 VGPRs at K = 2, 130 at K = 4;
//   * no workgroup barriers on the data path: a wave's own LDS traffic is ordered by the hardware, loads run D rows
//     ahead of the row being consumed, and the vector-memory counter is hand-counted: "row r has landed" is
//     s_waitcnt vmcnt(D - 1) -- only the D - 1 younger LOADS may be outstanding (the compiler cannot see the
//     dependence between an LDS-DMA and a later ds_read; stores of rows that do not exist go through an empty buffer
//     descriptor, so the instruction stream has no branches);
//   * recomputed halo: 7 K - 1 steps per CHUNK of several hundred rows instead of per 34-row tile; the 6 K shared
//     columns of neighbouring strips are read by neighbouring waves of one workgroup at about the same time
//     (optionally kept in step by one s_barrier per 7 rows) and hit in L1 / L2.
//
// Semantics are those of K consecutive launches of the reference driver from an even step: cells of an
// intermediate level outside the interior are 0 at odd levels and the source buffer's own halo value at even ones (under
// the Dirichlet option, K = 2, the input halo value); per accumulator the taps arrive in the same order as in the tile
// kernel, so results are bit-identical to it.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "rows_2d.h"

namespace lora {

// (a named type: the launcher below is compiled in four parts that hand it to each other, see LORA_STREAM_PART)
struct ArgsStream {
    const double *in;
    double *out;
    int ld, m, n;
    int row_begin, row_end;
    int strips;     // column strips
    int chunks;     // row chunks of the launch
    int rows;       // output rows per chunk: 7 groups - 7 K + 1
    int groups;     // groups of 7 steps per chunk
};

namespace {

constexpr int kSInW = 128;   // input columns per strip: one 16-byte piece per lane
constexpr int kBRow = 128;   // doubles of an intermediate row buffer; window reads of the lanes past a level's valid range
                             // run up to 6 doubles into whatever follows (the next row buffer, the next wave's ring,
                             // kLdsSlack at the end of the workgroup's allocation): never used, only has to be mapped
constexpr int kLdsSlack = 8;

// K applications per launch: every level shifts the lane -> column map by 3, so a strip yields 128 - 6 K output columns
// (116 for K = 2, 104 for K = 4) from its 128 input columns.
__host__ __device__ constexpr int stream_out_w(int K) { return kSInW - 6 * K; }
// Input-row ring slots per wave.  K = 2: seven, so that slot = step mod 7 is a compile-time constant of the 7-fold
// unrolled loop.  K = 4: ten (run-time slot index): the level-2 rows need the input row of 7 steps ago, see below.
__host__ __device__ constexpr int stream_slots(int K) { return K == 2 ? 7 : 10; }
__host__ __device__ constexpr int stream_wave_lds(int K) { return stream_slots(K) * kSInW + (K - 1) * kBRow; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One global -> LDS row piece with only the lanes of `mask` active (SHARE: a wave's quarter of the workgroup's row is
// 56 or 62 pieces of 16 bytes).  Written as one asm block that narrows EXEC around the instruction: the same thing as
// `if (lane < n) __builtin_amdgcn_global_load_lds(...)`, but that branch in every step made the register allocator
// spill 370 VGPRs of the rotating accumulators.  LDS address = M0 + 16 x lane.
__device__ __forceinline__ void dma_piece_masked(const double *gptr, unsigned lds_byte_addr, unsigned long long mask) {
    unsigned long long saved;
    unsigned saved_m0;  // M0 is the compiler's (a reserved register: it may not be named as clobbered) -- saved and restored here
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b32 %1, m0\n\t"
        "s_mov_b64 exec, %2\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, off\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved), "=&s"(saved_m0)
        : "s"(mask), "s"(lds_byte_addr), "v"(gptr)
        : "memory");
}

// Step r of a chunk whose first output row is i0 consumes input row i0 - 3 K + r and completes, for l = 1..K, the
// level-l row  i0 - 3 K - 4 l + 1 + r  (level l lags 3 rows -- its radius -- plus one step of hand-off behind level
// l - 1: a level's row is written to its LDS row buffer in one step and picked up by the next level in the following
// step, which lets one step run the K levels top down with ONE window in registers at a time).  Level K is the output.
//
// Cells of an intermediate level outside the interior are halo cells of the reference driver's buffers:
// never written, so 0 at odd levels (buffer 1) and the source buffer's own halo value at even levels (buffer 0) --
// or the source value at every level under the Dirichlet option (DIRI; K = 2 only).  The source value of a level-l
// cell is the same cell of the input row consumed 4 l - 1 steps earlier: still in the ring for (K = 2, l = 1, D <= 4)
// and for (K = 4, l = 2, ten slots, D <= 3).
//
// D = input rows in flight per wave; SYNC = 1: one s_barrier per 7 rows keeps the four strips of a workgroup in step
// (L2 hits on their shared columns), 2: one per row (L1 hits).
//
// SHARE: the four strips of a workgroup overlap by 6 K columns each and start at odd multiples of 8 bytes, so four private
// 1 KiB row pieces ask L2 for 36 lines of 128 bytes where the workgroup's 440 (476) distinct columns are 28 (31) -- and
// L2 -> L1 read requests are what bounds this kernel (64-86 in flight per CU whatever the occupancy).  In
// this mode the workgroup keeps ONE ring of whole rows: its window starts on a 128-byte boundary (padded column
// 4 OUTW g - 16), each wave fetches a quarter of it (112 / 124 columns), one s_barrier per row -- behind every wave's own
// vmcnt wait -- publishes the row, and every wave reads its window (and the halo source row) out of the shared row.
// The ring has one slot more than the slots a step touches (rows r, r - 7 or r - 3, r + D), because waves of one
// workgroup can be one barrier apart.
template <int EVAL, int K, int D, int SYNC, bool DIRI, bool PREFETCH, bool SHARE>
__global__ __launch_bounds__(256, (K == 2 ? 4 : 3)) void stencil2d_stream_kernel(const ArgsStream a, const Taps49 W, const LowRankTaps F) {
    constexpr int NS = SHARE ? (K == 2 ? 8 : 11) : stream_slots(K);
    constexpr int OUTW = stream_out_w(K);
    constexpr int SHIFT = 20 - 3 * K;                     // SHARE: ring column of the first strip's window
    constexpr int RW = SHARE ? (K == 2 ? 496 : 448) : kSInW;  // doubles per ring row (SHARE: the whole workgroup's)
    constexpr int RINGW = SHARE ? NS * RW : 4 * NS * RW;  // doubles of all rings of the workgroup
    static_assert(K == 2 || K == 4, "applications per launch");
    static_assert(D >= 2 && D <= 6 && (K == 2 || D <= 3) && (!DIRI || (K == 2 && D <= 4)), "rows in flight");
    static_assert(!SHARE || (SHIFT + 3 * OUTW + kSInW <= RW && RW % 8 == 0), "shared ring row");
    // K = 4: 4 x 13 KB + slack = 53,312 B -- three workgroups per CU need <= 53,760 B each (the 160 KB are handed out in
    // 1,280-byte granules: at 54,016 B only two were resident, PMC occupancy 1.74 of 3 waves per SIMD).  SHARE: 51,776 B.
    __shared__ __attribute__((aligned(128))) double lds[RINGW + 4 * (K - 1) * kBRow + kLdsSlack];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *const A = SHARE ? lds : lds + wv * NS * RW;         // ring: slot s = input row (step) s mod NS
    double *const B = lds + RINGW + wv * (K - 1) * kBRow;       // B + (l - 1) kBRow: the level-l row handed to level l + 1

    // wave -> (chunk, strip): consecutive waves take consecutive strips of one chunk; waves past the end of the
    // launch (last workgroup only; SHARE: past the end of a row of strips) run the same instruction stream with every
    // store switched off
    int chunk, strip;
    bool spare;
    const int blk = xcd_contiguous(blockIdx.x, gridDim.x);
    if (SHARE) {
        const int groups_x = (a.strips + 3) / 4;
        chunk = blk / groups_x;
        strip = (blk - chunk * groups_x) * 4 + wv;
        spare = strip >= a.strips;
    } else {
        const int total = a.strips * a.chunks;
        const int wlin = blk * 4 + wv;
        spare = wlin >= total;
        const int wl = spare ? total - 1 : wlin;
        chunk = wl / a.strips;
        strip = wl - chunk * a.strips;
    }
    const int i0 = a.row_begin + chunk * a.rows;  // first output row of the chunk (interior coordinates)
    const int j0 = strip * OUTW;                  // first output column of the strip
    const int row_hi = spare ? i0 : min(i0 + a.rows, a.row_end);

    // input: interior columns j0 - 3 K + 2 lane, +1 = padded columns j0 - 3 K + 4 + 2 lane (clamped into the padded
    // array: clamped pieces only feed intermediate cells outside the interior, which are forced below).  SHARE: this
    // wave's quarter of the workgroup's row, padded columns 4 OUTW g - 16 + (RW / 4) wv + 2 lane, lanes < RW / 8.
    const int gcol = SHARE ? min(max((strip - wv) * OUTW - 16 + (RW / 4) * wv + 2 * lane, 0), a.n + 6)
                           : min(max(j0 - 3 * K + 4 + 2 * lane, 0), a.n + 6);
    const double *const in_lane = a.in + gcol;
    auto issue = [&](int r, int slot) {
        const int pr = min(max(i0 - 3 * K + 4 + r, 0), a.m + 7);
        if constexpr (SHARE) {
            const unsigned dst = (unsigned) (unsigned long) (__attribute__((address_space(3))) void *) (A + slot * RW + (RW / 4) * wv);
            dma_piece_masked(in_lane + (size_t) pr * a.ld, dst, (1ull << (RW / 8)) - 1ull);
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (in_lane + (size_t) pr * a.ld),
                                             (__attribute__((address_space(3))) void *) (A + slot * RW), 16, 0, 0);
        }
    };
    // level-l columns of this lane: interior columns j0 - 3 K + 3 l + 2 lane, +1
    bool cin0[K], cin1[K];
#pragma unroll
    for (int l = 1; l < K; ++l) {
        const int jl = j0 - 3 * K + 3 * l + 2 * lane;
        cin0[l] = jl >= 0 && jl < a.n;
        cin1[l] = jl + 1 >= 0 && jl + 1 < a.n;
    }
    // stores: lanes 0 .. OUTW/2 - 1 write interior columns j0 + 2 lane, +1; the descriptor's range check drops the rest
    const unsigned store_off = lane < OUTW / 2 ? 16u * lane : 0x80000000u;
    const unsigned row_bytes = (unsigned) (max(min(OUTW, a.n - j0), 0) * 8);
    double *const out_strip = a.out + (j0 + 4);

    double pa[K][7], pb[K][7];  // rotating partial sums of the K levels, columns 2 lane / 2 lane + 1
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int k = 0; k < 7; ++k) pa[l][k] = pb[l][k] = 0.0;
#pragma unroll
    for (int l = 0; l < K - 1; ++l) *reinterpret_cast<d2 *>(B + l * kBRow + 2 * lane) = (d2){0.0, 0.0};

    // vector-memory stream: rows 0 .. D - 1 up front, then per step { one store, the load of row r + D }
#pragma unroll
    for (int k = 0; k < D; ++k) issue(k, k);

    const double *const winA = A + (SHARE ? SHIFT + OUTW * wv : 0) + 2 * lane;
    const double *const winB = B + 2 * lane;

    // sl = ring slot of input row r (= phase when NS == 7)
    auto step = [&](const int r, const int sl, auto phase_tag) {
        constexpr int P = decltype(phase_tag)::value;  // r mod 7
        // "input row r has landed": at most the D - 1 younger loads may be outstanding.  (Loads complete in order among
        // themselves.  Counting the interleaved stores as well -- vmcnt(2 (D - 1)) -- assumes a younger store never
        // completes before an older LDS-DMA load; the same assumption in kernels_3d_bf16_mfma.hip gave a wrong plane
        // about once in 400 runs.  With D >= 3 this wait does not stall on store acknowledgements: the loads of the
        // rows in between landed long ago, so the store of the previous step may still be in flight.)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        if (SYNC == 2 || SHARE) {  // SHARE: every wave's quarter of row r has landed once all have passed here
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        auto wrap = [](int x) { return x >= NS ? x - NS : x; };
        // Window of the NEXT level to run is fetched while the current one computes (PREFETCH: +16 VGPRs, hides the LDS
        // latency inside the wave; every window of a step was written in the previous step, so the order is free)
        d2 wnext[4];
        auto fetch = [&](int l) {  // window of level l: its source row buffer, or the ring slot of input row r
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wnext[q] = l >= 2 ? *reinterpret_cast<const d2 *>(winB + (l - 2) * kBRow + 2 * q)
                                  : *reinterpret_cast<const d2 *>(winA + sl * RW + 2 * q);
        };
        if (PREFETCH) fetch(K);
        // ---- levels K .. 2: the level-(l-1) row completed in the previous step, from its row buffer ----
#pragma unroll
        for (int l = K; l >= 2; --l) {
            const int R = (P + 7 * K - (l - 1)) % 7;  // (r - (l - 1)) mod 7: the rotation this level is at
            double win[8];
            {
                if (!PREFETCH) fetch(l);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    win[2 * q] = wnext[q].x;
                    win[2 * q + 1] = wnext[q].y;
                }
                if (PREFETCH) fetch(l - 1);
            }
            // R is a compile-time constant after unrolling; dispatch it to the template parameter
            auto run = [&](auto rot_tag) {
                constexpr int ROT = decltype(rot_tag)::value;
                apply_row<EVAL, 7, ROT>(6, win, pa[l - 1], pb[l - 1], W, F);
                d2 v;
                v.x = pa[l - 1][ROT];
                v.y = pb[l - 1][ROT];
                pa[l - 1][ROT] = 0.0;
                pb[l - 1][ROT] = 0.0;
                const int row = i0 - 3 * K - 4 * l + 1 + r;  // interior row of the completed level-l row
                if (l == K) {
                    const bool live = row >= i0 && row < row_hi;
                    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
                        out_strip + (size_t) (max(row, 0) + 4) * a.ld, 0, live ? row_bytes : 0u, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst, store_off, 0, 0);
                } else {
                    const bool row_in = row >= 0 && row < a.m;
                    const bool in0 = row_in && cin0[l], in1 = row_in && cin1[l];
                    if ((l & 1) == 0 || DIRI) {
                        // halo cell of an even level: the source buffer's own value = input row r - 4 l + 1, columns
                        // 2 lane + 3 l, +1 of its ring slot (an aligned 16-byte piece for even l)
                        const d2 h = *reinterpret_cast<const d2 *>(winA + wrap(sl + NS - (4 * l - 1) % NS) * RW + 3 * l);
                        v.x = in0 ? v.x : h.x;
                        v.y = in1 ? v.y : h.y;
                    } else {
                        v.x = in0 ? v.x : 0.0;
                        v.y = in1 ? v.y : 0.0;
                    }
                    *reinterpret_cast<d2 *>(B + (l - 1) * kBRow + 2 * lane) = v;
                }
            };
            switch (R) {
                case 0: run(std::integral_constant<int, 0>{}); break;
                case 1: run(std::integral_constant<int, 1>{}); break;
                case 2: run(std::integral_constant<int, 2>{}); break;
                case 3: run(std::integral_constant<int, 3>{}); break;
                case 4: run(std::integral_constant<int, 4>{}); break;
                case 5: run(std::integral_constant<int, 5>{}); break;
                default: run(std::integral_constant<int, 6>{}); break;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- level 1: input row r ----
        {
            double win[8];
            {
                if (!PREFETCH) fetch(1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    win[2 * q] = wnext[q].x;
                    win[2 * q + 1] = wnext[q].y;
                }
            }
            apply_row<EVAL, 7, P>(6, win, pa[0], pb[0], W, F);
            d2 v;
            v.x = pa[0][P];
            v.y = pb[0][P];
            pa[0][P] = 0.0;
            pb[0][P] = 0.0;
            const int row = i0 - 3 * K - 3 + r;
            const bool row_in = row >= 0 && row < a.m;
            const bool in0 = row_in && cin0[1], in1 = row_in && cin1[1];
            if (DIRI) {
                // source value: input row r - 3, columns 2 lane + 3, +1 (8-byte aligned only)
                const double *cell = winA + wrap(sl + NS - 3) * RW + 3;
                const double h0 = cell[0], h1 = cell[1];
                v.x = in0 ? v.x : h0;
                v.y = in1 ? v.y : h1;
            } else {
                v.x = in0 ? v.x : 0.0;
                v.y = in1 ? v.y : 0.0;
            }
            *reinterpret_cast<d2 *>(B + 2 * lane) = v;
        }
        // The compiler does not know that the LDS-DMA below overwrites a ring slot: with D at its maximum that is the
        // very slot the halo-cell reads above took their values from, so nothing of this step may sink below it.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        issue(r + D, wrap(sl + D));
        __builtin_amdgcn_sched_barrier(0);
    };

    int sl0 = 0;  // ring slot of the first step of the group (always 0 when NS == 7)
    for (int g = 0; g < a.groups; ++g) {
        if (SYNC == 1 && !SHARE) __builtin_amdgcn_s_barrier();
        const int r0 = 7 * g;
        auto slot = [&](int k) { return NS == 7 ? k : (sl0 + k >= NS ? sl0 + k - NS : sl0 + k); };
        step(r0 + 0, slot(0), std::integral_constant<int, 0>{});
        step(r0 + 1, slot(1), std::integral_constant<int, 1>{});
        step(r0 + 2, slot(2), std::integral_constant<int, 2>{});
        step(r0 + 3, slot(3), std::integral_constant<int, 3>{});
        step(r0 + 4, slot(4), std::integral_constant<int, 4>{});
        step(r0 + 5, slot(5), std::integral_constant<int, 5>{});
        step(r0 + 6, slot(6), std::integral_constant<int, 6>{});
        if (NS != 7) sl0 = sl0 + 7 >= NS ? sl0 + 7 - NS : sl0 + 7;
    }
    // drain: the last D loads target this wave's LDS, which the next workgroup on this CU may own
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EVAL, int K, int D, int SYNC, bool DIRI, bool PREFETCH, bool SHARE = false>
hipError_t launch_stream_t(const Plan &p, const ArgsStream &a, hipStream_t s) {
    Taps49 w;
    for (int k = 0; k < 49; ++k) w.w[k] = p.w[k];
    LowRankTaps f{};
    for (int t = 0; t < 3; ++t)
        for (int e = 0; e < 7; ++e) {
            f.u[t][e] = p.lowrank.u[t][e];
            f.v[t][e] = p.lowrank.v[t][e];
        }
    f.rc = p.lowrank_rc;
    if (EVAL == EVAL_NEST)
        for (int k = 0; k < 4; ++k) {
            f.u[0][k] = p.nest_g[k];
            f.v[0][k] = p.nest_a[k];
        }
    const long waves = (long) a.strips * a.chunks;
    const long nblocks = SHARE ? (long) ((a.strips + 3) / 4) * a.chunks : (waves + 3) / 4;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((stencil2d_stream_kernel<EVAL, K, D, SYNC, DIRI, PREFETCH, SHARE>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w, f);
    return hipGetLastError();
}

template <int EVAL, int K>
hipError_t launch_stream_e(const Plan &p, const ArgsStream &a, bool dirichlet, hipStream_t s) {
    const int sync = p.stream_sync;
    // window prefetch (option stream_prefetch, K = 4): 1039 vs 1040 us per launch -- LDS latency is not what bounds it
    if (p.stream_share) {  // one ring of whole rows per workgroup (D fixed: 3 rows in flight; 4 under Dirichlet)
        if constexpr (K == 2) {
            if (dirichlet) return launch_stream_t<EVAL, 2, 4, 1, true, false, true>(p, a, s);
            return launch_stream_t<EVAL, 2, 4, 1, false, false, true>(p, a, s);
        } else {
            return launch_stream_t<EVAL, 4, 3, 1, false, false, true>(p, a, s);
        }
    }
#define LORA_STREAM_GO(DD, DIRI)                                                                                  \
    return sync == 2   ? launch_stream_t<EVAL, K, DD, 2, DIRI, false>(p, a, s)                                     \
           : sync == 1 ? (K == 4 && p.stream_prefetch ? launch_stream_t<EVAL, K, DD, 1, DIRI, true>(p, a, s)       \
                                                      : launch_stream_t<EVAL, K, DD, 1, DIRI, false>(p, a, s))     \
                       : launch_stream_t<EVAL, K, DD, 0, DIRI, false>(p, a, s);
    if constexpr (K == 2) {
        if (dirichlet) {  // re-reads the input row three steps back: at most 4 rows in flight
            LORA_STREAM_GO(4, true)
        }
        if (p.stream_depth == 2) { LORA_STREAM_GO(2, false) }
        if (p.stream_depth == 3) { LORA_STREAM_GO(3, false) }
        if (p.stream_depth == 6) { LORA_STREAM_GO(6, false) }
        LORA_STREAM_GO(4, false)
    } else {
        if (p.stream_depth == 2) { LORA_STREAM_GO(2, false) }
        LORA_STREAM_GO(3, false)
    }
#undef LORA_STREAM_GO
}

}  // namespace

hipError_t stream_part0(int eval, int K, const Plan &p, const ArgsStream &a, bool dirichlet, hipStream_t s);
hipError_t stream_part1(int eval, int K, const Plan &p, const ArgsStream &a, bool dirichlet, hipStream_t s);
hipError_t stream_part2(int eval, int K, const Plan &p, const ArgsStream &a, bool dirichlet, hipStream_t s);
hipError_t stream_part3(int eval, int K, const Plan &p, const ArgsStream &a, bool dirichlet, hipStream_t s);

#ifndef LORA_STREAM_PART
#define LORA_STREAM_PART 0
#endif
#define LORA_STREAM_PAIR(NAME, E0, E1)                                                                              \
    hipError_t NAME(int eval, int K, const Plan &p, const ArgsStream &a, bool dirichlet, hipStream_t s) {          \
        if (eval == E0) return K == 2 ? launch_stream_e<E0, 2>(p, a, dirichlet, s) : launch_stream_e<E0, 4>(p, a, dirichlet, s); \
        return K == 2 ? launch_stream_e<E1, 2>(p, a, dirichlet, s) : launch_stream_e<E1, 4>(p, a, dirichlet, s);  \
    }
#if LORA_STREAM_PART == 0
LORA_STREAM_PAIR(stream_part0, EVAL_NEST, TAPS2D_BOX)
#elif LORA_STREAM_PART == 1
LORA_STREAM_PAIR(stream_part1, EVAL_LR_DIAMOND, EVAL_LR_PYRAMID)
#elif LORA_STREAM_PART == 2
LORA_STREAM_PAIR(stream_part2, EVAL_LR_PYRAMID_SYM, EVAL_LR_PYRAMID_SYM_GAP)
#else
LORA_STREAM_PAIR(stream_part3, TAPS2D_DIAMOND, TAPS2D_STAR)
#endif
#undef LORA_STREAM_PAIR

#if LORA_STREAM_PART == 0
// Rows per chunk.  A chunk costs 7 K - 1 extra steps (recomputed rows), so chunks should be long; but the launch should
// also fill every CU evenly: with j workgroups on most CUs and j + 1 on a few, the few set the pace (measured,
// star2d1r 16384^2, K = 4: 435 rows = 5.9 workgroups per CU 939 GStencils/s, 582 rows = 4.5 per CU 884; box2d3r 8192^2:
// 330 rows = 1.9 per CU 663, 295 rows = 2.2 per CU 535).  So: the fewest whole workgroups-per-CU j whose chunks are no
// longer than a limit around which the rate is flat (a few rounds of workgroups also even out the speed of waves).
int stream_rows_per_chunk(const Plan &p, int K, int rows_total, int strips) {
    const int lag = 7 * K - 1;  // steps of a chunk beyond its output rows
    auto fit = [&](long rows) {
        long g = (rows + lag + 6) / 7;
        if (g < K + 1) g = K + 1;
        return (int) (7 * g - lag);
    };
    if (p.stream_rows > 0) return fit(p.stream_rows);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int limit = K == 2 ? 640 : 448;
    int best = fit(rows_total);
    // at least 4 (K = 2: memory-bound, wants its full 16 waves per CU) or 2 (K = 4) workgroups per CU
    for (int j = (K == 2 ? 4 : 2); j <= 256; ++j) {
        const long chunks = 4L * cus * j / strips;  // 4 waves (strips) per workgroup
        if (chunks < 1) continue;
        best = fit((rows_total + chunks - 1) / chunks);
        if (best <= limit) break;
    }
    return best;
}

int stream_strip_width(int K) { return stream_out_w(K); }

// K = 2 or 4 applications in one launch over interior rows [begin, end)
hipError_t launch_2d_stream(const Plan &p, int K, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    if (K != 2 && K != 4) return hipErrorInvalidValue;
    const bool dirichlet = p.boundary == LORA_BC_DIRICHLET;
    if (K == 4 && dirichlet) return hipErrorNotSupported;  // the driver keeps K = 2 under the Dirichlet option
    ArgsStream a;
    a.in = in;
    a.out = out;
    a.m = p.dims[0];
    a.n = p.dims[1];
    a.ld = a.n + 8;
    a.row_begin = begin;
    a.row_end = end;
    a.strips = (a.n + stream_out_w(K) - 1) / stream_out_w(K);
    a.rows = stream_rows_per_chunk(p, K, end - begin, a.strips);
    a.groups = (a.rows + 7 * K - 1) / 7;
    a.chunks = (end - begin + a.rows - 1) / a.rows;
    // the instantiations are spread over four translation units (this file compiled with LORA_STREAM_PART = 0 .. 3:
    // ~200 unrolled kernels took 4.5 minutes in one): each part serves two of the eight tap evaluations
    switch (p.fused_eval) {
        case EVAL_NEST:
            return stream_part0(EVAL_NEST, K, p, a, dirichlet, s);
        case EVAL_LR_DIAMOND:
            return stream_part1(EVAL_LR_DIAMOND, K, p, a, dirichlet, s);
        case EVAL_LR_PYRAMID:
            return stream_part1(EVAL_LR_PYRAMID, K, p, a, dirichlet, s);
        case EVAL_LR_PYRAMID_SYM:
            return stream_part2(EVAL_LR_PYRAMID_SYM, K, p, a, dirichlet, s);
        case EVAL_LR_PYRAMID_SYM_GAP:
            return stream_part2(EVAL_LR_PYRAMID_SYM_GAP, K, p, a, dirichlet, s);
        default:
            break;
    }
    switch (p.tapset) {
        case TAPS2D_DIAMOND:
            return stream_part3(TAPS2D_DIAMOND, K, p, a, dirichlet, s);
        case TAPS2D_STAR:
            return stream_part3(TAPS2D_STAR, K, p, a, dirichlet, s);
        default:
            return stream_part0(TAPS2D_BOX, K, p, a, dirichlet, s);
    }
}

const char *kernel_name_2d_stream(const Plan &) { return "stencil2d_stream_kernel"; }

#endif  // LORA_STREAM_PART == 0

}  // namespace lora
