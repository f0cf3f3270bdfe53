// kernels_2d_stream.hip -- TWO kernel applications of a radius-3 2D stencil per launch, row-streaming form.
//
// The tile kernel of kernels_2d_fused.hip holds a whole 46 x 136 input window in LDS per workgroup: 50 KB, three
// workgroups per CU, and inside a workgroup strictly serial phases (window load -> barrier -> application 1 ->
// barrier -> application 2 + stores).  It moves only the compulsory HBM bytes but at 4.6-5.0 TB/s (PMC, profiles/),
// below the 5.6 TB/s the one-sweep kernel streams at: while a workgroup computes it has no loads in flight, 18 % of
// its level-1 rows and 35 % of its window rows are recomputed / re-read halo, and a third of its vector instructions
// are tile bookkeeping (clamped addresses, guards).
//
// Here every WAVE is autonomous and walks DOWN a column strip (the time-step loop of the reference driver,
// 2d/gpu.cu:544-546, two steps per pass; kernel replaced: 2d/gpu.cu:181-273):
//   * a strip is 116 output columns = 122 intermediate columns = 128 input columns: exactly one 16-byte piece per
//     lane and row, so a row of the strip is ONE global_load_lds_dwordx4 (1 KiB, global -> LDS, no staging
//     registers) and one buffer_store_dwordx4;
//   * the scatter form of apply_row needs each input row once: LDS only serves as the cross-lane window exchange of
//     the row being consumed (4 x ds_read_b128 per row and level, as in the tile kernel), so a wave owns a ring of
//     seven 1 KiB input-row slots (rows in flight) plus one intermediate row: 8.3 KB per wave, 16 waves per CU;
//   * the seven partially summed rows of each level live in rotating register files (the loop is unrolled by 7 so
//     that the rotation is a compile-time renaming, apply_row<.., ROT>);
//   * no workgroup barriers on the data path: a wave's own LDS traffic is ordered by the hardware, loads run D rows
//     ahead of the row being consumed, and the vector-memory counter is hand-counted -- every step issues exactly one
//     store and one load, so "row r has landed" is the constant s_waitcnt vmcnt(2 (D - 1)) (stores of rows that do
//     not exist go through an empty buffer descriptor and are dropped by its range check; the compiler cannot see
//     the dependence between an LDS-DMA and a later ds_read);
//   * recomputed halo: 12 input rows and 6 intermediate rows per CHUNK of several hundred rows instead of per
//     34-row tile; the 12 shared columns of neighbouring strips are read by neighbouring waves of one workgroup at
//     about the same time (optionally kept in step by one s_barrier per 7 rows) and hit in L1 / L2.
//
// Semantics are those of the tile kernel (two consecutive launches of the reference driver from an even step,
// SURVEY B1/B2): the intermediate level's cells outside the interior are 0 (or, under the Dirichlet option, keep the
// input halo value); per accumulator the taps arrive in the same order, so results are bit-identical to it.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "rows_2d.h"

namespace lora {

namespace {

constexpr int kSOutW = 116;  // output columns per strip (58 lanes x 2)
constexpr int kSInW = 128;   // input columns per strip: one 16-byte piece per lane
constexpr int kSlots = 7;    // input-row ring slots per wave (= the unroll factor, so slot = phase)
constexpr int kBRow = 136;   // doubles of the intermediate row buffer (lanes past 60 read slack)
constexpr int kWaveLds = kSlots * kSInW + kBRow;  // doubles of LDS per wave

struct ArgsStream {
    const double *in;
    double *out;
    int ld, m, n;
    int row_begin, row_end;
    int strips;     // column strips
    int chunks;     // row chunks of the launch
    int rows;       // output rows per chunk: 7 groups - 6
    int groups;     // main-loop groups of 7 steps per chunk
    int dirichlet;  // intermediate cells outside the interior keep the input halo value instead of 0
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// D = input rows in flight per wave (1..6; <= 4 under the Dirichlet option, which re-reads the row 3 steps back);
// SYNC = one s_barrier per 7 rows keeps the four strips of a workgroup in step (L1 / L2 hits on their shared columns).
template <int EVAL, int D, int SYNC, bool DIRI>
__global__ __launch_bounds__(256, 4) void stencil2d_stream2_kernel(const ArgsStream a, const Taps49 W, const LowRankTaps F) {
    static_assert(D >= 1 && D <= 6, "rows in flight");
    __shared__ __attribute__((aligned(16))) double lds[4 * kWaveLds];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *const A = lds + wv * kWaveLds;  // ring: slot s = input row (step) s mod 7
    double *const B = A + kSlots * kSInW;   // the intermediate row being handed from level 1 to level 2

    // wave -> (chunk, strip): consecutive waves take consecutive strips of one chunk; waves past the end of the
    // launch (last workgroup only) run the same instruction stream on the last strip with every store switched off
    const int total = a.strips * a.chunks;
    const int wlin = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wv;
    const bool spare = wlin >= total;
    const int wl = spare ? total - 1 : wlin;
    const int chunk = wl / a.strips, strip = wl - chunk * a.strips;
    const int i0 = a.row_begin + chunk * a.rows;  // first output row of the chunk (interior coordinates)
    const int j0 = strip * kSOutW;                // first output column of the strip
    const int row_hi = spare ? i0 : min(i0 + a.rows, a.row_end);

    // step r consumes input row i0 - 6 + r = padded row i0 - 2 + r, padded columns j0 - 2 + 2 lane, +1 (clamped into
    // the padded array: clamped pieces only feed intermediate cells outside the interior, which are forced below)
    const int gcol = min(max(j0 - 2 + 2 * lane, 0), a.n + 6);
    const double *const in_lane = a.in + gcol;
    auto issue = [&](int r, int slot) {
        const int pr = min(max(i0 - 2 + r, 0), a.m + 7);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (in_lane + (size_t) pr * a.ld),
                                         (__attribute__((address_space(3))) void *) (A + slot * kSInW), 16, 0, 0);
    };
    // intermediate columns of this lane: interior columns j0 - 3 + 2 lane, +1
    const int jm = j0 - 3 + 2 * lane;
    const bool c0_in = jm >= 0 && jm < a.n;
    const bool c1_in = jm + 1 >= 0 && jm + 1 < a.n;
    // stores: lanes 0..57 write interior columns j0 + 2 lane, +1; the descriptor's range check drops the rest
    const unsigned store_off = lane < kSOutW / 2 ? 16u * lane : 0x80000000u;
    const unsigned row_bytes = (unsigned) (min(kSOutW, a.n - j0) * 8);
    double *const out_strip = a.out + (j0 + 4);

    double p1a[7], p1b[7], p2a[7], p2b[7];  // rotating partial sums of the two levels, columns 2 lane / 2 lane + 1
#pragma unroll
    for (int k = 0; k < 7; ++k) p1a[k] = p1b[k] = p2a[k] = p2b[k] = 0.0;

    // vector-memory stream: L(0), then per step exactly { store, L(r + D) }: D - 1 dummy { store, load } pairs up front
    // make "younger than L(r)" the same 2 (D - 1) operations at every step
    const __amdgpu_buffer_rsrc_t nowhere = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);
    issue(0, 0);
#pragma unroll
    for (int k = 1; k < D; ++k) {
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, nowhere, 0, 0, 0);
        issue(k, k);
    }

    const double *const winA = A + 2 * lane;
    const double *const winB = B + 2 * lane;

    auto step = [&](const int r, auto phase_tag, auto full_tag) {
        constexpr int P = decltype(phase_tag)::value;     // r mod 7 = ring slot of input row r
        constexpr bool FULL = decltype(full_tag)::value;  // level 2 runs from step 7 on
        constexpr int Q = (P + 6) % 7;                    // (r - 1) mod 7
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
        // ---- level 1: input row r into the intermediate rows r .. r+6 (tap row 6 - k of row r + k) ----
        {
            d2 wa[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wa[q] = *reinterpret_cast<const d2 *>(winA + P * kSInW + 2 * q);
            double win[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                win[2 * q] = wa[q].x;
                win[2 * q + 1] = wa[q].y;
            }
            apply_row<EVAL, 7, P>(6, win, p1a, p1b, W, F);
        }
        // intermediate row r (interior row i0 - 9 + r) is complete; level 2 picks up row r - 1 from B first, then this
        // row takes its place (a wave's LDS operations execute in order)
        d2 wb[4];
        if constexpr (FULL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wb[q] = *reinterpret_cast<const d2 *>(winB + 2 * q);
        }
        {
            const int im = i0 - 9 + r;
            const bool row_in = im >= 0 && im < a.m;
            d2 v;
            v.x = (row_in && c0_in) ? p1a[P] : 0.0;
            v.y = (row_in && c1_in) ? p1b[P] : 0.0;
            if constexpr (DIRI) {
                // the cell's own input value: input row r - 3 (slot (P + 4) mod 7, still resident for D <= 4)
                const double *cell = winA + ((P + 4) % 7) * kSInW + 3;
                const double h0 = cell[0], h1 = cell[1];
                v.x = (row_in && c0_in) ? v.x : h0;
                v.y = (row_in && c1_in) ? v.y : h1;
            }
            *reinterpret_cast<d2 *>(B + 2 * lane) = v;
            p1a[P] = 0.0;
            p1b[P] = 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- level 2: intermediate row r - 1 into the output rows; output row r - 1 (interior row i0 - 13 + r) is
        //      then complete ----
        if constexpr (FULL) {
            double win[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                win[2 * q] = wb[q].x;
                win[2 * q + 1] = wb[q].y;
            }
            apply_row<EVAL, 7, Q>(6, win, p2a, p2b, W, F);
            const int ro = i0 - 13 + r;
            const bool live = ro >= i0 && ro < row_hi;
            const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
                out_strip + (size_t) (max(ro, 0) + 4) * a.ld, 0, live ? row_bytes : 0u, 0x00020000);
            d2 v;
            v.x = p2a[Q];
            v.y = p2b[Q];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst, store_off, 0, 0);
            p2a[Q] = 0.0;
            p2b[Q] = 0.0;
        } else {
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, nowhere, 0, 0, 0);
        }
        issue(r + D, (P + D) % 7);
        __builtin_amdgcn_sched_barrier(0);
    };

    using T = std::true_type;
    using Fa = std::false_type;
#define LORA_STEP7(r0, FULLT)                                  \
    step((r0) + 0, std::integral_constant<int, 0>{}, FULLT{}); \
    step((r0) + 1, std::integral_constant<int, 1>{}, FULLT{}); \
    step((r0) + 2, std::integral_constant<int, 2>{}, FULLT{}); \
    step((r0) + 3, std::integral_constant<int, 3>{}, FULLT{}); \
    step((r0) + 4, std::integral_constant<int, 4>{}, FULLT{}); \
    step((r0) + 5, std::integral_constant<int, 5>{}, FULLT{}); \
    step((r0) + 6, std::integral_constant<int, 6>{}, FULLT{});
    LORA_STEP7(0, Fa)
    for (int g = 0; g < a.groups; ++g) {
        if (SYNC) __builtin_amdgcn_s_barrier();
        const int r0 = 7 + 7 * g;
        LORA_STEP7(r0, T)
    }
#undef LORA_STEP7
    // drain: the last D loads target this wave's LDS, which the next workgroup on this CU may own
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EVAL, int D, int SYNC, bool DIRI>
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
    const long waves = (long) a.strips * a.chunks;
    const long nblocks = (waves + 3) / 4;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((stencil2d_stream2_kernel<EVAL, D, SYNC, DIRI>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w, f);
    return hipGetLastError();
}

template <int EVAL>
hipError_t launch_stream_e(const Plan &p, const ArgsStream &a, hipStream_t s) {
    const bool sync = p.stream_sync != 0;
    if (a.dirichlet)  // re-reads the input row three steps back: at most 4 rows in flight
        return sync ? launch_stream_t<EVAL, 4, 1, true>(p, a, s) : launch_stream_t<EVAL, 4, 0, true>(p, a, s);
#define LORA_STREAM_D(DD)     \
    if (p.stream_depth == DD) \
        return sync ? launch_stream_t<EVAL, DD, 1, false>(p, a, s) : launch_stream_t<EVAL, DD, 0, false>(p, a, s);
    LORA_STREAM_D(2)
    LORA_STREAM_D(3)
    LORA_STREAM_D(5)
    LORA_STREAM_D(6)
#undef LORA_STREAM_D
    return sync ? launch_stream_t<EVAL, 4, 1, false>(p, a, s) : launch_stream_t<EVAL, 4, 0, false>(p, a, s);
}

}  // namespace

// Rows per chunk: as long as possible (the 12 + 6 recomputed rows are per chunk) while the launch still has a whole
// number of "rounds" of the chip's resident waves -- 256 CUs x 16 waves -- so that no round runs half empty.
int stream2_rows_per_chunk(const Plan &p, int rows_total, int strips) {
    if (p.stream_rows > 0) {
        int k = (p.stream_rows + 6 + 6) / 7;
        return 7 * (k < 2 ? 2 : k) - 6;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const long resident = 16L * cus;
    int best = 0;
    for (int round = 1; round <= 64; ++round) {
        long chunks = resident * round / strips;
        if (chunks < 1) continue;
        long rows = (rows_total + chunks - 1) / chunks;
        long k = (rows + 6 + 6) / 7;
        if (k < 2) k = 2;
        rows = 7 * k - 6;
        best = (int) rows;
        if (rows <= 640) break;
    }
    if (best <= 0) best = 8;
    return best;
}

hipError_t launch_2d_stream2(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    ArgsStream a;
    a.in = in;
    a.out = out;
    a.m = p.dims[0];
    a.n = p.dims[1];
    a.ld = a.n + 8;
    a.row_begin = begin;
    a.row_end = end;
    a.strips = (a.n + kSOutW - 1) / kSOutW;
    a.rows = stream2_rows_per_chunk(p, end - begin, a.strips);
    a.groups = (a.rows + 6) / 7;
    a.chunks = (end - begin + a.rows - 1) / a.rows;
    a.dirichlet = p.boundary == LORA_BC_DIRICHLET;
    switch (p.fused_eval) {
        case EVAL_LR_DIAMOND:
            return launch_stream_e<EVAL_LR_DIAMOND>(p, a, s);
        case EVAL_LR_PYRAMID:
            return launch_stream_e<EVAL_LR_PYRAMID>(p, a, s);
        case EVAL_LR_PYRAMID_SYM:
            return launch_stream_e<EVAL_LR_PYRAMID_SYM>(p, a, s);
        case EVAL_LR_PYRAMID_SYM_GAP:
            return launch_stream_e<EVAL_LR_PYRAMID_SYM_GAP>(p, a, s);
        default:
            break;
    }
    switch (p.tapset) {
        case TAPS2D_DIAMOND:
            return launch_stream_e<TAPS2D_DIAMOND>(p, a, s);
        case TAPS2D_STAR:
            return launch_stream_e<TAPS2D_STAR>(p, a, s);
        default:
            return launch_stream_e<TAPS2D_BOX>(p, a, s);
    }
}

const char *kernel_name_2d_stream2(const Plan &) { return "stencil2d_stream2_kernel"; }

}  // namespace lora
