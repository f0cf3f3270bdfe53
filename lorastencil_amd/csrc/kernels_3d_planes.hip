// kernels_3d_planes.hip -- K = 2 or 3 applications of a 3D radius-1 stencil per launch (fp64): plane streaming.
//
// kernels_3d_fused.hip (two applications, 30 x 60 output tiles, z-chunks of 16 planes, one register-staged plane of
// prefetch) moves 1.24 x its compulsory HBM bytes at ~4.9 TB/s -- it is HBM-bound on the bytes it really moves
// (profiles/r02_star3d1r_pmc.json), so the only way up is fewer bytes per application.  This kernel
//   * fuses THREE applications per launch: the grid is read once and written once per three sweeps;
//   * uses 512-thread workgroups (62 x 60 / 60 x 60 output tiles from a 66 x 68 input window: 1.21 / 1.25 x instead of
//     1.28 x) and z-chunks of 64 planes for K = 3 (2 K re-read planes per chunk), whole rounds of long chunks for K = 2;
//   * streams input planes global -> LDS with global_load_lds (no staging registers) into a ring of NS plane slots, the
//     next plane(s) in flight while the current one is consumed; waits are counted in LOADS only (a younger store may
//     complete before an older LDS-DMA load: kernels_2d_stream.hip);
//   * needs K workgroup barriers per plane for K = 3 (raw s_barrier after lgkmcnt(0) -- __syncthreads would also drain
//     vmcnt and with it the prefetch); K = 2, and K = 3 on request, run "pipelined": every level consumes what was
//     published one plane step earlier, from the other of two buffers, with ONE barrier per plane.
//
// Levels.  Level 0 is the input, level l the result of l applications; level l lives in LDS tile T_l (row stride 68
// doubles), its cells produced by the same lane -> cell map at every level: strip sid = 2 wave + lane / 32 owns rows
// 4 sid .. 4 sid + 3, lane column pair cl = lane % 32.  A lane reads a window of 6 rows x 4 columns around its cells
// from T_(l-1) (three aligned ds_read_b128 per row, conflict-free), scatters it into three rotating
// accumulator sets (planes_3d.h) and publishes the completed plane.  Tile coordinates:
//   T_0 (i, c)  <->  interior row I - K + i, column J - 4 + c      (padded column J + c: every 16-byte piece aligned)
//   T_l cell of (sid, r, cl)  <->  interior row I - K + l + 4 sid + r, columns J - 4 + base_l + 2 cl, + 1
// Level l + 1 reads T_l at columns 2 cl .. 2 cl + 5; level l is published at column 2 cl + pub_l, pub_l = 2 (the next
// level keeps the lane's columns) or 0 (the next level's columns move one lane to the right), so that every pair
// stays 16-byte aligned while each level needs only ONE more column per side: base_1 = 2, K = 2: pub_1 = 2, output
// lanes 1 .. 30; K = 3: pub_1 = 2, pub_2 = 0, output lanes 0 .. 29 -- 60 output columns either way.  Rows shift by one
// per level: MH = 8 NW level-1 rows, MH - 2 (K - 1) output rows.
//
// Semantics: the K launches of the reference driver that start at global step s0: level l is global
// level s0 + l, whose buffer's halo is the caller's halo when s0 + l is even and 0 when it is odd.  `halo_src` points
// at a buffer that carries the caller's halo (buffer 0), `parity` = s0 mod 2; with the Dirichlet option every level
// keeps the caller's halo.  The input's own halo is whatever `in` holds: with K = 3 the driver (capi.cpp) runs the
// launches on the reference's natural buffer state (buffer 0: caller's halo, buffer 1: zeros), so launch k reads the
// halo the reference's step 3 k + 1 would read, and no halo copies or scratch grid are needed.  Taps are applied in the
// single-sweep kernel's order at every level: the result is bit-identical to K single sweeps -- except for exactly
// separable box taps (TAPS3D_SEP: the reference's own), which run as x / y / z passes (planes_3d.h, 9-10 instead of 27
// multiply-adds per point): identical while every partial sum is an exact integer, ~1e-16 relative per sweep otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include <cstdio>
#include <cstdlib>

#include "device_common.h"
#include "planes_3d.h"

namespace lora {

namespace {

constexpr int kRY = 4;          // rows per strip
constexpr int kLanesX = 32;     // lanes per strip
constexpr int kRowW = 68;       // LDS row stride of every level (doubles) = staged input columns
constexpr int kPieces = kRowW / 2;  // 16-byte pieces per staged row
constexpr int kOutW = 60;       // output columns per tile

struct ArgsS3 {
    const double *in;
    double *out;
    const double *halo_src;  // a buffer whose halo is the caller's (read for halo cells of even global levels)
    int h, m, n;
    int ld;
    long plane;
    int z_begin, z_end;
    int zc;
    int tiles_x, tiles_y;
    int parity;  // global step the launch starts at, mod 2
    unsigned long long *dbg;  // -DLORA_DIAGNOSTICS: per-phase cycle sums of a few waves
    int ablate;  // -DLORA_DIAGNOSTICS timing experiments (results wrong): 1 no barriers, 2 no taps, 3 no stores, 4 no loads, 5 no publish
};

__device__ __forceinline__ void lds_barrier(bool skip = false) {
    if (skip)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A 16-byte global load the compiler does not track: issued and waited for inside one asm statement.  A plain load
// anywhere in the plane loop makes the compiler's own waitcnt insertion put s_waitcnt vmcnt(0) in front of every use
// of (and every LDS read into) the registers it may still be writing -- on the common path too, where it drains the
// LDS-DMA prefetch and the output stores at every level (the first version of this kernel: 2-3 x slower).  Only waves
// on the grid's rim get here.
__device__ __forceinline__ d2 load_halo_pair(const double *p) {
    d2 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// Four of them behind ONE wait, in ONE statement: the loads' destination registers are outputs of the statement that also
// waits for them, so the compiler never sees a register whose load is still in flight (issued and waited for in two
// statements, it may copy, re-materialise or spill such a register in between and keep the stale value -- the register-
// resident 3D kernel's first version computed wrong planes that way at 256 VGPRs).  Every lane loads; lanes that need
// nothing pass a harmless address and ignore the result.
__device__ __forceinline__ void load_halo_pairs4(d2 (&v)[4], const double *p0, const double *p1, const double *p2, const double *p3) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\t"
        "global_load_dwordx4 %1, %5, off\n\t"
        "global_load_dwordx4 %2, %6, off\n\t"
        "global_load_dwordx4 %3, %7, off\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_loads() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// window of row j of a strip (taps use elements 1 .. 4); the star reads only its own pair in the two rows above /
// below the strip's cells
template <int TAPSET>
__device__ __forceinline__ void read_window(const double *row, bool inner, double (&win)[6]) {
    const d2 c = *reinterpret_cast<const d2 *>(row + 2);
    win[2] = c.x;
    win[3] = c.y;
    if (TAPSET != TAPS3D_STAR || inner) {
        // Three aligned 16-byte reads (4 LDS cycles each, conflict-free).  Two 8-byte reads of elements 1 and 4 become
        // one ds_read2_b64: 8 cycles and, at a lane stride of 16 bytes, 2-way bank conflicts (30 % of this kernel's LDS
        // cycles in its first version).
        const d2 l = *reinterpret_cast<const d2 *>(row);
        const d2 r = *reinterpret_cast<const d2 *>(row + 4);
        win[0] = l.x;
        win[1] = l.y;
        win[4] = r.x;
        win[5] = r.y;
    } else {
        win[0] = win[1] = win[4] = win[5] = 0.0;
    }
}

template <int K>
struct Geo {
    // pub[l]: LDS column offset level l (1 .. K - 1) is published at; base[l]: tile column of lane 0's pair at level l
    static constexpr int pub(int l) { return K == 3 && l == 2 ? 0 : 2; }
    static constexpr int base(int l) {
        int b = 2;
        for (int q = 1; q < l; ++q) b += pub(q) == 0 ? 2 : 0;
        return b;
    }
    static constexpr int out_lane0 = K == 2 ? 1 : 0;  // first lane whose level-K pair is complete
};

template <int TAPSET, int K, int NW, int NS, bool PIPE, bool DIRICHLET>
__global__ __launch_bounds__(NW * 64, 2) void stencil3d_planes_kernel(const ArgsS3 a, const Taps27 W) {
    constexpr int STRIPS = 2 * NW;
    constexpr int MH = STRIPS * kRY;              // level-1 rows
    constexpr int IH = MH + 2;                    // input rows
    constexpr int OH = MH - 2 * (K - 1);          // output rows
    constexpr int NPIECE = IH * kPieces;          // 16-byte pieces per input plane
    constexpr int NINST = (NPIECE + 63) / 64;     // wave-wide LDS-DMA instructions per plane (1 KiB each)
    constexpr int CNT = (NINST + NW - 1) / NW;    // per wave: the first REM waves issue CNT, the others CNT - 1
    constexpr int REM = NINST - (CNT - 1) * NW;
    constexpr int SLOT = NPIECE * 2;              // doubles per ring slot (the last instruction overlaps the one before)
    constexpr int TILE = MH * kRowW;              // doubles per level tile (the last strip reads 2 rows past it: the
                                                  // next tile's first rows or the 2-row pad at the end -- never used)
    constexpr int YOUNG = (K == 2 || PIPE) ? NS - 2 : NS - 1;  // planes issued after the one a step waits for
    constexpr int NB = PIPE ? 2 : 1;                 // buffers per published level
    static_assert(K == 2 || K == 3, "two or three applications per launch");
    static_assert(K == 3 || PIPE, "two applications: one published level, so the one-barrier form with two buffers");
    static_assert(Geo<K>::base(K) + 2 * Geo<K>::out_lane0 == 4, "output columns start at tile column 4");

    // One dynamic LDS block.  (Two static arrays were tried: the compiler then knows array A is what the LDS-DMA writes
    // and puts s_waitcnt vmcnt(0) in front of every read of it -- correct without the hand-counted waits below, but it
    // drains the prefetch.  Through the dynamic block it leaves the ordering to the explicit waits.)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *const A = reinterpret_cast<double *>(smem);  // NS input plane slots
    double *const B = A + NS * SLOT;                     // levels 1 .. K - 1 (PIPE: two buffers each), 2 pad rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sid = wv * 2 + lane / kLanesX;
    const int cl = lane % kLanesX;
    const bool big = wv < REM;  // this wave issues CNT pieces per plane

    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int per_chunk = a.tiles_x * a.tiles_y;
    const int chunk = lin / per_chunk;
    const int rem = lin - chunk * per_chunk;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;  // first output plane of the chunk
    const int I = ty * OH;
    const int J = tx * kOutW;
    const int zc = min(a.zc, a.z_end - k0);
    // input planes: interior k0 - K .. k0 + zc + K - 1; PIPE: level l runs l - 1 steps behind, K - 1 more steps drain it
    const int nplanes = zc + 2 * K + (PIPE ? K - 1 : 0);

    // LDS-DMA pieces of this lane: instruction q = wv + t NW covers pieces 64 q .. 64 q + 63 of the plane image, the last
    // one the image's last 64 pieces (overlapping its predecessor with the same data: the image is not a multiple of
    // 1 KiB).  Source pieces are clamped into the padded array; clamped pieces only feed cells outside the interior.
    int goff[CNT];
#pragma unroll
    for (int t = 0; t < CNT; ++t) {
        const int g = min((wv + t * NW) * 64, NPIECE - 64) + lane;
        const int r = g / kPieces;
        const int c = g - r * kPieces;
        const int gr = min(max(I - K + 2 + r, 0), a.m + 3);
        const int gc = min(J + 2 * c, a.n + 6);
        goff[t] = gr * a.ld + gc;
    }
    auto issue = [&](int p) {
        const double *src = a.in + (long) min(max(k0 - K + 1 + p, 0), a.h + 1) * a.plane;
        double *slot = A + (p % NS) * SLOT;
#pragma unroll
        for (int t = 0; t < CNT; ++t) {
            if ((t < CNT - 1 || big) && !(LORA_ABLATE(a) & 8))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + goff[t]),
                                                 (__attribute__((address_space(3))) void *) (slot + 2 * min((wv + t * NW) * 64, NPIECE - 64)),
                                                 16, 0, 0);
        }
    };
    // "the plane issued YOUNG planes ago has landed": only the younger LOADS may be outstanding
    auto wait_plane = [&]() {
        if (big)
            wait_loads<YOUNG * CNT>();
        else
            wait_loads<YOUNG *(CNT - 1)>();
    };

    double x0[K][3][kRY], x1[K][3][kRY];  // rotating partial sums of the K levels
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < kRY; ++r) x0[l][s][r] = x1[l][s][r] = 0.0;

    // geometry of this lane at level l: interior row I - K + l + 4 sid + r, columns J - 4 + base_l + 2 cl (+1)
    const int strip_off = (sid * kRY) * kRowW + 2 * cl;
    const bool lvl_halo[3] = {false, DIRICHLET || ((a.parity + 1) & 1) == 0, DIRICHLET || ((a.parity + 2) & 1) == 0};
    const int colo = J - 4 + Geo<K>::base(K) + 2 * cl;  // output columns
    const bool colo_ok = cl >= Geo<K>::out_lane0 && cl < Geo<K>::out_lane0 + kOutW / 2 && colo < a.n;
    const int rowo = I + sid * kRY;
    double *const out_col = a.out + (long) (rowo + 2) * a.ld + (colo + 4);
    // Loop-invariant, wave-uniform shortcuts (a wave issues one instruction every few cycles, and at two waves per SIMD
    // the length of its instruction stream is what bounds this kernel: the per-lane range checks below cost more
    // scalar instructions per plane than the taps cost vector ones).  xy_all[l]: every cell this wave owns at level l
    // lies inside the interior in x and y -- then a plane inside the z range is published without any per-lane check;
    // store_all: every lane stores all of its four output rows.
    bool xy_all[K];
#pragma unroll
    for (int l = 1; l < K; ++l) {
        const int row = I - K + l + sid * kRY;
        const int col = J - 4 + Geo<K>::base(l) + 2 * cl;
        const bool lane_in = col >= 0 && col < a.n && row >= 0 && row + kRY - 1 < a.m;
        xy_all[l] = __builtin_amdgcn_ballot_w64(!lane_in) == 0;
    }
    const bool store_all =
        __builtin_amdgcn_ballot_w64(!(colo_ok && sid * kRY + kRY - 1 < OH && rowo + kRY - 1 < a.m)) == 0;

#pragma unroll
    for (int p = 0; p < NS; ++p) issue(p);
    if (big)
        wait_loads<(NS - 1) * CNT>();
    else
        wait_loads<(NS - 1) * (CNT - 1)>();
    lds_barrier();

#ifdef LORA_DIAGNOSTICS
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    const bool dbg_on = a.dbg != nullptr && (blockIdx.x == 0 || blockIdx.x == 97) && lane == 0;
#define S3_T(i)                                                   \
    {                                                             \
        const unsigned long long tnow = __builtin_amdgcn_s_memtime(); \
        tacc[i] += tnow - tlast;                                  \
        tlast = tnow;                                             \
    }
#else
#define S3_T(i)
#endif
    // level L (compile-time) of step p: scatter plane p - LAG (L - 1) of level L - 1, publish / store the plane of level L
    // below it.  LAG = 1: each level consumes what the previous one published in this step, behind a barrier.  PIPE
    // (LAG = 2): it consumes what was published in the PREVIOUS step -- the K levels of a step are independent, one
    // barrier per step, two buffers per published level.
    constexpr int LAG = PIPE ? 2 : 1;
    auto level = [&](int p, auto level_tag, auto phase_tag) {
        constexpr int L = decltype(level_tag)::value;
        constexpr int PH = decltype(phase_tag)::value;  // (p - LAG (L - 1)) mod 3
        const double *strip =
            (L == 1 ? A + (p % NS) * SLOT : B + ((L - 2) * NB + (PIPE ? ((p + 1) & 1) : 0)) * TILE) + strip_off;
        // all six window rows are fetched before the first tap: one LDS round trip per level, not one per row (two
        // waves per SIMD do not hide six)
        double win[kRY + 2][6];
#pragma unroll
        for (int j = 0; j < kRY + 2; ++j) {
            if ((LORA_ABLATE(a) & 32)) {
#pragma unroll
                for (int e = 0; e < 6; ++e) win[j][e] = x1[L - 1][e % 3][j % kRY];
                continue;
            }
            read_window<TAPSET>(strip + j * kRowW, j >= 1 && j <= kRY, win[j]);
        }
        // (elements 0 and 5 are never used by a tap; kept alive up to here, or the register allocator overlaps the
        // windows and serialises the reads behind lgkmcnt(0) every third one)
#pragma unroll
        for (int j = 1; j <= kRY; ++j) asm volatile("" ::"v"(win[j][0]), "v"(win[j][5]));
        __builtin_amdgcn_sched_barrier(0);
#ifdef LORA_DIAGNOSTICS
        if (L == 2 && a.dbg) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            S3_T(8)
        }
#endif
#pragma unroll
        for (int j = 0; j < kRY + 2; ++j) {
            if ((LORA_ABLATE(a) & 2)) {
                x0[L - 1][0][0] += win[j][1] + win[j][2] + win[j][3] + win[j][4];
                continue;
            }
            if constexpr (TAPSET != TAPS3D_SEP) scatter_row<TAPSET, kRY, PH, true>(x0[L - 1], x1[L - 1], win[j], j, W);
        }
        if constexpr (TAPSET == TAPS3D_SEP) scatter_plane_sep<kRY, PH, true>(x0[L - 1], x1[L - 1], win, W);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < kRY; ++r) asm volatile("" : "+v"(x0[L - 1][s][r]), "+v"(x1[L - 1][s][r]));
#ifdef LORA_DIAGNOSTICS
        if (L == 2 && a.dbg) S3_T(9)
#endif
        // the plane of level L that became complete: index p - LAG (L - 1) - 1, slot (PH + 1) % 3
        constexpr int s = (PH + 1) % 3;
        if constexpr (L < K) {
            const int z = k0 - K + p - LAG * (L - 1) - 1;
            const int row = I - K + L + sid * kRY;
            const int col = J - 4 + Geo<K>::base(L) + 2 * cl;
            const bool zc_in = z >= 0 && z < a.h && col >= 0 && col < a.n;  // n is even: a pair is on one side
            double *dst = B + ((L - 1) * NB + (PIPE ? (p & 1) : 0)) * TILE + strip_off + Geo<K>::pub(L);
            // Cells outside the interior hold 0 (odd global levels) or the caller's halo value (even global levels, every
            // level under the fixed boundary), read from `halo_src`.  All of that sits in a branch of its own, taken by
            // the few waves on the grid's rim: the common path publishes its sums as they are.  (The compiler waits
            // for vmcnt(0) before it uses a loaded register; on the common path that wait drained the whole prefetch
            // stream -- LDS-DMA loads and output stores -- at every publish.)
            if (!(z >= 0 && z < a.h && xy_all[L])) {
                // the four rows' halo loads are issued together and waited for once: a rim workgroup's plane step would
                // otherwise carry four serial memory round trips per level, and with one round of long z-chunks the
                // slowest (rim) workgroups set the launch time
                d2 hv[kRY];
                bool need[kRY];
                const double *hp[kRY];
#pragma unroll
                for (int r = 0; r < kRY; ++r) {
                    const bool in = zc_in && row + r >= 0 && row + r < a.m;
                    const int pz = z + 1, pr = row + r + 2, pc = col + 4;
                    // cells beyond the padded array feed no valid output
                    need[r] = !in && lvl_halo[L] && pz >= 0 && pz <= a.h + 1 && pr >= 0 && pr <= a.m + 3 && pc >= 0 &&
                              pc + 1 <= a.n + 7;
                    hp[r] = need[r] ? a.halo_src + (long) pz * a.plane + (long) pr * a.ld + pc : a.halo_src;
                }
                static_assert(kRY == 4, "four halo loads per level");
                if (lvl_halo[L]) {
                    load_halo_pairs4(hv, hp[0], hp[1], hp[2], hp[3]);
#pragma unroll
                    for (int r = 0; r < kRY; ++r) hv[r] = need[r] ? hv[r] : (d2){0.0, 0.0};
                } else {
#pragma unroll
                    for (int r = 0; r < kRY; ++r) hv[r] = (d2){0.0, 0.0};
                }
#pragma unroll
                for (int r = 0; r < kRY; ++r) {
                    const bool in = zc_in && row + r >= 0 && row + r < a.m;
                    d2 v;
                    v.x = in ? x0[L - 1][s][r] : hv[r].x;
                    v.y = in ? x1[L - 1][s][r] : hv[r].y;
                    if (!(LORA_ABLATE(a) & 16)) *reinterpret_cast<d2 *>(dst + r * kRowW) = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < kRY; ++r) {
                    d2 v;
                    v.x = x0[L - 1][s][r];
                    v.y = x1[L - 1][s][r];
                    if (!(LORA_ABLATE(a) & 16)) *reinterpret_cast<d2 *>(dst + r * kRowW) = v;
                }
            }
        } else {
            const int o = p - K - LAG * (K - 1) - 1;
            if (o >= 0 && o < zc && !(LORA_ABLATE(a) & 4)) {
                double *dst = out_col + (long) (k0 + o + 1) * a.plane;
                if (store_all) {
#pragma unroll
                    for (int r = 0; r < kRY; ++r) {
                        d2 v;
                        v.x = x0[L - 1][s][r];
                        v.y = x1[L - 1][s][r];
                        *reinterpret_cast<d2 *>(dst + (long) r * a.ld) = v;
                    }
                } else if (colo_ok) {
#pragma unroll
                    for (int r = 0; r < kRY; ++r) {
                        if (sid * kRY + r < OH && rowo + r < a.m) {
                            d2 v;
                            v.x = x0[L - 1][s][r];
                            v.y = x1[L - 1][s][r];
                            *reinterpret_cast<d2 *>(dst + (long) r * a.ld) = v;
                        }
                    }
                }
            }
        }
    };

    auto step = [&](int p, auto phase_tag) {
        constexpr int PH0 = decltype(phase_tag)::value;  // p mod 3
        if constexpr (PIPE) {
            level(p, std::integral_constant<int, 1>{}, std::integral_constant<int, PH0>{});
            level(p, std::integral_constant<int, 2>{}, std::integral_constant<int, (PH0 + 1) % 3>{});
            if constexpr (K == 3) level(p, std::integral_constant<int, 3>{}, std::integral_constant<int, (PH0 + 2) % 3>{});
            wait_plane();
            lds_barrier((LORA_ABLATE(a) & 1));
            issue(p + NS);
        } else {
            S3_T(7)
            level(p, std::integral_constant<int, 1>{}, std::integral_constant<int, PH0>{});
            S3_T(0)
            // the last barrier of the step also publishes the NEXT input plane (every wave waits for its own pieces
            // first); after the first barrier every wave is done with input plane p, whose slot takes plane p + NS
            if constexpr (K == 2) wait_plane();
            lds_barrier((LORA_ABLATE(a) & 1));
            S3_T(1)
            issue(p + NS);
            S3_T(2)
            level(p, std::integral_constant<int, 2>{}, std::integral_constant<int, (PH0 + 2) % 3>{});
            S3_T(3)
            if constexpr (K == 3) {
                wait_plane();
                S3_T(4)
                lds_barrier((LORA_ABLATE(a) & 1));
                S3_T(5)
                level(p, std::integral_constant<int, 3>{}, std::integral_constant<int, (PH0 + 1) % 3>{});
                S3_T(6)
            }
        }
    };

    for (int p = 0; p < nplanes; p += 3) {
        step(p, std::integral_constant<int, 0>{});
        if (p + 1 < nplanes) step(p + 1, std::integral_constant<int, 1>{});
        if (p + 2 < nplanes) step(p + 2, std::integral_constant<int, 2>{});
    }
    wait_loads<0>();  // no LDS-DMA may be in flight when the wave ends
#ifdef LORA_DIAGNOSTICS
    if (dbg_on)
        for (int i = 0; i < 12; ++i) a.dbg[((blockIdx.x ? 1 : 0) * NW + wv) * 12 + i] = tacc[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The same computation WITHOUT workgroup barriers ("async", plan option stream3_async).
//
// With one workgroup per CU at two waves per SIMD, K barriers per plane put the eight waves in lockstep: all read their
// windows, then all compute, then all publish -- the LDS and the vector pipe are used one after the other.
// But a wave depends on very little of what the others do:
//   * its level-(l+1) window reads rows 8 w .. 8 w + 9 of level l: its own eight rows and the first TWO rows its lower
//     neighbour (wave w + 1) publishes;
//   * it may overwrite its own rows of level l for the next plane once its UPPER neighbour (wave w - 1) has read them.
// So every wave gets (a) a private ring of input plane pieces -- the ten input rows it reads, loaded by itself with
// LDS-DMA, two of them also by its neighbour: no cross-wave dependence on input at all, its own vmcnt is enough -- and
// (b) two counters per published level in LDS: pub[l][w] = planes of level l wave w has published, rd[l][w] = planes of
// level l it has finished reading.  Wave w waits for pub[l][w + 1] before it reads level l and for rd[l][w - 1] before
// it overwrites it.  A wave's LDS operations execute in order, so "data, then counter" needs no fence; waits go from
// wave w to w + 1 (same plane) or to w - 1 (previous plane), so they cannot form a cycle; and every wait loop is
// bounded -- a lost update would end in wrong numbers (caught by the parity tests), never in a hung GPU.
// The levels are the barrier version's, instruction for instruction: results are bit-identical.
__device__ __forceinline__ void flag_set(int *f, int v) {
    asm volatile("" ::: "memory");
    *reinterpret_cast<volatile int *>(f) = v;
}

__device__ __forceinline__ void flag_wait(const int *f, int target) {
    for (int spin = 0; spin < (1 << 18); ++spin) {  // ~20 ms: never reached unless a counter update is lost
        const int v = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(f));
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

template <int TAPSET, int K, int NW, bool DIRICHLET>
__global__ __launch_bounds__(NW * 64, 2) void stencil3d_planes_async_kernel(const ArgsS3 a, const Taps27 W) {
    constexpr int NS = 2;                         // input plane slots per wave
    constexpr int STRIPS = 2 * NW;
    constexpr int MH = STRIPS * kRY;              // level-1 rows
    constexpr int OH = MH - 2 * (K - 1);          // output rows
    constexpr int AROWS = 2 * kRY + 2;            // input rows one wave reads
    constexpr int APIECE = AROWS * kPieces;       // 340 pieces of 16 bytes
    constexpr int AINST = (APIECE + 63) / 64;     // 6 LDS-DMA instructions per wave and plane
    constexpr int ASLOT = APIECE * 2;             // doubles per wave slot
    constexpr int TILE = MH * kRowW;
    static_assert(K == 2 || K == 3, "two or three applications per launch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *const A = reinterpret_cast<double *>(smem);          // NW x NS private slots
    double *const B = A + NW * NS * ASLOT;                       // levels 1 .. K - 1, then 2 pad rows
    int *const flags = reinterpret_cast<int *>(B + (K - 1) * TILE + 2 * kRowW);  // pub[K - 1][NW], rd[K - 1][NW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sid = wv * 2 + lane / kLanesX;
    const int cl = lane % kLanesX;

    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int per_chunk = a.tiles_x * a.tiles_y;
    const int chunk = lin / per_chunk;
    const int rem = lin - chunk * per_chunk;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;
    const int I = ty * OH;
    const int J = tx * kOutW;
    const int zc = min(a.zc, a.z_end - k0);
    const int nplanes = zc + 2 * K;

    if (tid < 2 * (K - 1) * NW) flags[tid] = 0;
    __syncthreads();  // the only workgroup barrier: counters start at 0
    int *const pub_mine = flags + wv;             // + (l - 1) * NW
    int *const rd_mine = flags + (K - 1) * NW + wv;

    // this wave's input rows: tile rows 8 w .. 8 w + 9 <-> padded rows I - K + 2 + 8 w + r
    int goff[AINST];
#pragma unroll
    for (int t = 0; t < AINST; ++t) {
        const int g = min(t * 64, APIECE - 64) + lane;
        const int r = g / kPieces;
        const int c = g - r * kPieces;
        const int gr = min(max(I - K + 2 + 2 * kRY * wv + r, 0), a.m + 3);
        const int gc = min(J + 2 * c, a.n + 6);
        goff[t] = gr * a.ld + gc;
    }
    double *const Aw = A + wv * NS * ASLOT;
    auto issue = [&](int p) {
        const double *src = a.in + (long) min(max(k0 - K + 1 + p, 0), a.h + 1) * a.plane;
        double *slot = Aw + (p % NS) * ASLOT;
#pragma unroll
        for (int t = 0; t < AINST; ++t)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + goff[t]),
                                             (__attribute__((address_space(3))) void *) (slot + 2 * min(t * 64, APIECE - 64)), 16, 0,
                                             0);
    };

    double x0[K][3][kRY], x1[K][3][kRY];
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < kRY; ++r) x0[l][s][r] = x1[l][s][r] = 0.0;

    const int strip_off = (sid * kRY) * kRowW + 2 * cl;                    // in a level tile
    const int strip_offA = ((lane / kLanesX) * kRY) * kRowW + 2 * cl;      // in this wave's input slot
    const bool lvl_halo[3] = {false, DIRICHLET || ((a.parity + 1) & 1) == 0, DIRICHLET || ((a.parity + 2) & 1) == 0};
    const int colo = J - 4 + Geo<K>::base(K) + 2 * cl;
    const bool colo_ok = cl >= Geo<K>::out_lane0 && cl < Geo<K>::out_lane0 + kOutW / 2 && colo < a.n;
    const int rowo = I + sid * kRY;
    double *const out_col = a.out + (long) (rowo + 2) * a.ld + (colo + 4);

#pragma unroll
    for (int p = 0; p < NS; ++p) issue(p);

    auto level = [&](int p, auto level_tag, auto phase_tag) {
        constexpr int L = decltype(level_tag)::value;
        constexpr int PH = decltype(phase_tag)::value;  // (p - (L - 1)) mod 3
        if constexpr (L == 1) {
            wait_loads<(NS - 1) * AINST>();  // this wave's pieces of plane p have landed (younger loads: plane p + 1)
        } else {
            // level L - 1 of this plane step: the lower neighbour's first two rows must be there
            if (wv + 1 < NW) flag_wait(pub_mine + (L - 2) * NW + 1, p + 1);
        }
        const double *strip = L == 1 ? Aw + (p % NS) * ASLOT + strip_offA : B + (L - 2) * TILE + strip_off;
        double win[kRY + 2][6];
#pragma unroll
        for (int j = 0; j < kRY + 2; ++j) read_window<TAPSET>(strip + j * kRowW, j >= 1 && j <= kRY, win[j]);
#pragma unroll
        for (int j = 1; j <= kRY; ++j) asm volatile("" ::"v"(win[j][0]), "v"(win[j][5]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the window is in registers
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (L == 1) {
            issue(p + NS);  // the slot is free again
        } else {
            flag_set(rd_mine + (L - 2) * NW, p + 1);  // the upper neighbour's rows of level L - 1 may be overwritten
        }
        if constexpr (TAPSET == TAPS3D_SEP) {
            scatter_plane_sep<kRY, PH, true>(x0[L - 1], x1[L - 1], win, W);
        } else {
#pragma unroll
            for (int j = 0; j < kRY + 2; ++j) scatter_row<TAPSET, kRY, PH, true>(x0[L - 1], x1[L - 1], win[j], j, W);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < kRY; ++r) asm volatile("" : "+v"(x0[L - 1][s][r]), "+v"(x1[L - 1][s][r]));
        constexpr int s = (PH + 1) % 3;
        if constexpr (L < K) {
            const int z = k0 - K + p - (L - 1) - 1;
            const int row = I - K + L + sid * kRY;
            const int col = J - 4 + Geo<K>::base(L) + 2 * cl;
            const bool zc_in = z >= 0 && z < a.h && col >= 0 && col < a.n;
            double *dst = B + (L - 1) * TILE + strip_off + Geo<K>::pub(L);
            // the upper neighbour has read what the previous plane step left in this wave's first two rows
            if (wv > 0) flag_wait(rd_mine + (L - 1) * NW - 1, p);
            bool rim = false;
#pragma unroll
            for (int r = 0; r < kRY; ++r) rim |= !(zc_in && row + r >= 0 && row + r < a.m);
            if (__builtin_amdgcn_ballot_w64(rim) != 0) {
#pragma unroll
                for (int r = 0; r < kRY; ++r) {
                    const bool in = zc_in && row + r >= 0 && row + r < a.m;
                    d2 v;
                    v.x = in ? x0[L - 1][s][r] : 0.0;
                    v.y = in ? x1[L - 1][s][r] : 0.0;
                    if (!in && lvl_halo[L]) {
                        const int pz = z + 1, pr = row + r + 2, pc = col + 4;
                        if (pz >= 0 && pz <= a.h + 1 && pr >= 0 && pr <= a.m + 3 && pc >= 0 && pc + 1 <= a.n + 7)
                            v = load_halo_pair(a.halo_src + (long) pz * a.plane + (long) pr * a.ld + pc);
                    }
                    *reinterpret_cast<d2 *>(dst + r * kRowW) = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < kRY; ++r) {
                    d2 v;
                    v.x = x0[L - 1][s][r];
                    v.y = x1[L - 1][s][r];
                    *reinterpret_cast<d2 *>(dst + r * kRowW) = v;
                }
            }
            flag_set(pub_mine + (L - 1) * NW, p + 1);  // after the rows, in this wave's LDS order
        } else {
            const int o = p - 2 * K;
            if (o >= 0 && o < zc && colo_ok) {
                double *dst = out_col + (long) (k0 + o + 1) * a.plane;
#pragma unroll
                for (int r = 0; r < kRY; ++r) {
                    if (sid * kRY + r < OH && rowo + r < a.m) {
                        d2 v;
                        v.x = x0[L - 1][s][r];
                        v.y = x1[L - 1][s][r];
                        *reinterpret_cast<d2 *>(dst + (long) r * a.ld) = v;
                    }
                }
            }
        }
    };

    auto step = [&](int p, auto phase_tag) {
        constexpr int PH0 = decltype(phase_tag)::value;
        level(p, std::integral_constant<int, 1>{}, std::integral_constant<int, PH0>{});
        level(p, std::integral_constant<int, 2>{}, std::integral_constant<int, (PH0 + 2) % 3>{});
        if constexpr (K == 3) level(p, std::integral_constant<int, 3>{}, std::integral_constant<int, (PH0 + 1) % 3>{});
    };
    for (int p = 0; p < nplanes; p += 3) {
        step(p, std::integral_constant<int, 0>{});
        if (p + 1 < nplanes) step(p + 1, std::integral_constant<int, 1>{});
        if (p + 2 < nplanes) step(p + 2, std::integral_constant<int, 2>{});
    }
    wait_loads<0>();  // no LDS-DMA may be in flight when the wave ends
}

template <int K, int NW>
constexpr size_t stream3_async_lds_bytes() {
    return (size_t) (NW * 2 * (2 * kRY + 2) * kRowW + (K - 1) * 2 * NW * kRY * kRowW + 2 * kRowW) * sizeof(double) +
           2 * (K - 1) * NW * sizeof(int);
}

template <int K, int NW, int NS, bool PIPE>
constexpr size_t stream3_lds_bytes() {
    constexpr int MH = 2 * NW * kRY;
    return (size_t) (NS * (MH + 2) * kRowW + (PIPE ? 2 : 1) * (K - 1) * MH * kRowW + 2 * kRowW) * sizeof(double);
}

// z-chunk length.  Two applications: the longest chunks (2 K re-read planes each) that still give every workgroup slot
// of the chip work, in as few whole rounds of workgroups as possible.  Three applications: 64 planes -- several rounds
// of workgroups at different depths (tools/zc_sweep.sh, star3d1r 512^3 / 768^3, GStencils/s: 32 planes 617 / 723, 64:
// 615 / 729, 86: 572 / 718, 171: 598 / 646, 256: 447 / 651): workgroups on the grid's rim publish through the slower
// halo path, and in a single round of long chunks they alone set the launch time.
int stream3_chunk(const Plan &p, int K, int NW, bool pipe, long tiles, int depth) {
    if (p.fused_z_chunk > 0) return p.fused_z_chunk;
    if (K == 3) return depth < 64 ? depth : 64;
    const long slots = 256L * (NW == 4 ? 2 : 1);  // workgroups the chip holds at once
    double best = -1.0;
    int best_zc = depth;
    for (int rounds = 1; rounds <= 4; ++rounds) {
        long chunks = rounds * slots / tiles;
        if (chunks < 1) chunks = 1;
        if (chunks > depth) chunks = depth;
        const int zc = (int) ((depth + chunks - 1) / chunks);
        chunks = (depth + zc - 1) / zc;
        const long wgs = chunks * tiles;
        const long r = (wgs + slots - 1) / slots;
        // useful share of the occupied slot-time: slot filling x planes that are not re-read
        const double score = (double) wgs / (double) (r * slots) * (double) zc / (double) (zc + 2 * K + (pipe ? K - 1 : 0));
        if (score > best + 0.02) {
            best = score;
            best_zc = zc;
        }
    }
    return best_zc;
}

template <int TAPSET, int K, int NW, int NS, bool PIPE>
hipError_t launch_stream3(const Plan &p, const double *in, double *out, const double *halo_src, int parity, int begin,
                          int end, hipStream_t s) {
    constexpr int OH = 2 * NW * kRY - 2 * (K - 1);
    ArgsS3 a;
    a.in = in;
    a.out = out;
    a.halo_src = halo_src;
    a.parity = parity & 1;
    a.ablate = p.ablate;
    a.dbg = nullptr;
#ifdef LORA_DIAGNOSTICS
    static unsigned long long *dbg = nullptr;
    if (getenv("LORA_S3_DEBUG")) {
        if (!dbg) (void) hipMalloc(&dbg, 2 * 8 * 12 * sizeof(unsigned long long));
        (void) hipMemsetAsync(dbg, 0, 2 * 8 * 12 * sizeof(unsigned long long), s);
        a.dbg = dbg;
    }
#endif
    a.h = p.dims[0];
    a.m = p.dims[1];
    a.n = p.dims[2];
    a.ld = a.n + 8;
    a.plane = (long) (a.m + 4) * (a.n + 8);
    if (a.plane >= (1L << 31)) return hipErrorInvalidValue;  // 32-bit in-plane offsets
    a.z_begin = begin;
    a.z_end = end;
    a.tiles_x = (a.n + kOutW - 1) / kOutW;
    a.tiles_y = (a.m + OH - 1) / OH;
    a.zc = stream3_chunk(p, K, NW, PIPE, (long) a.tiles_x * a.tiles_y, end - begin);
    const long chunks = ((long) end - begin + a.zc - 1) / a.zc;
    const long nblocks = chunks * a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    Taps27 w;
    for (int k = 0; k < 27; ++k) w.w[k] = p.w[k];
    if (TAPSET == TAPS3D_SEP)
        for (int k = 0; k < 9; ++k) w.w[k] = p.sep64[k];
    constexpr bool ASYNC = NS == 0;  // NS = 0 selects the barrier-free kernel
    constexpr size_t lds = ASYNC ? stream3_async_lds_bytes<K, NW>() : stream3_lds_bytes<K, NW, NS ? NS : 2, PIPE>();
    static_assert(lds <= (NW == 4 ? 80 : 160) * 1024, "LDS budget");
    auto go = [&](auto kernel) -> hipError_t {
        // per KERNEL and device: more than 64 KiB of dynamic LDS has to be asked for once.  (Keyed by the kernel's address:
        // the two boundary variants have the same function-pointer type and therefore share this lambda's statics.)
        static std::mutex mu;
        static std::vector<std::pair<const void *, int>> prepared;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        {
            const std::pair<const void *, int> key(reinterpret_cast<const void *>(kernel), dev);
            std::lock_guard<std::mutex> lock(mu);
            if (std::find(prepared.begin(), prepared.end(), key) == prepared.end()) {
                const hipError_t e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
                if (e != hipSuccess) return e;
                prepared.push_back(key);
            }
        }
        hipLaunchKernelGGL(kernel, dim3((unsigned) nblocks), dim3(NW * 64), lds, s, a, w);
#ifdef LORA_DIAGNOSTICS
        if (a.dbg) {
            unsigned long long h[2 * 8 * 12];
            (void) hipStreamSynchronize(s);
            (void) hipMemcpy(h, a.dbg, sizeof h, hipMemcpyDeviceToHost);
            const long steps = a.zc + 2 * K;
            for (int b = 0; b < 2; ++b)
                for (int wq = 0; wq < NW; wq += NW - 1) {
                    std::fprintf(stderr, "s3dbg K=%d NW=%d block=%d wave=%d cycles/step:", K, NW, b ? 97 : 0, wq);
                    for (int i = 0; i < 12; ++i) std::fprintf(stderr, " %llu", h[(b * NW + wq) * 12 + i] / steps);
                    std::fprintf(stderr, "\n");
                }
        }
#endif
        return hipGetLastError();
    };
    if constexpr (ASYNC) {
        if (p.boundary == LORA_BC_DIRICHLET) return go(stencil3d_planes_async_kernel<TAPSET, K, NW, true>);
        return go(stencil3d_planes_async_kernel<TAPSET, K, NW, false>);
    } else {
        if (p.boundary == LORA_BC_DIRICHLET) return go(stencil3d_planes_kernel<TAPSET, K, NW, NS, PIPE, true>);
        return go(stencil3d_planes_kernel<TAPSET, K, NW, NS, PIPE, false>);
    }
}

}  // namespace

// workgroup shapes, bounded by 160 KiB of LDS per CU (80 KiB for two 4-wave workgroups per CU): 8 waves (one workgroup
// per CU) or 4 (two), always two input plane slots; the pipelined three-level form fits with 6 waves only.  (Deeper
// rings -- 7 waves x 3 slots, 6 x 4 -- were built and measured within 3 % of these; they are gone again.)
int stream3_slots(int, int, int, int) { return 2; }

int stream3_waves(const Plan &p, int K, int pipe) {
    int req = p.stream3_waves;
    if (req == 0) {
        // automatic: 8 waves (one workgroup per CU); two 4-wave workgroups per CU for the separable box on grids below
        // ~3e7 points (tools/rule3d.sh, box3d1r 128^3 / 192^3 / 256^3: 169 / 351 / 454 against 155 / 323 / 438 GStencils/s)
        const double npts = (double) p.dims[0] * p.dims[1] * p.dims[2];
        req = (p.sep64_valid && npts < 3.0e7) ? 4 : 8;
    }
    const int nw = req == 4 ? 4 : 8;
    return (pipe && K == 3) ? 6 : nw;  // the pipelined three-level form fits with 6 waves only
}

// K applications starting at global step `parity` (mod 2); `halo_src`: a buffer that carries the caller's halo.
hipError_t launch_3d_stream(const Plan &p, int K, const double *in, double *out, const double *halo_src, int parity,
                            int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    // Two applications always run in the one-barrier form: with a single published level, the barrier between "publish"
    // and "consume" does not keep a fast wave from publishing the next plane over rows a slow neighbour still reads
    // (the three-level form has its second barrier in between) -- found by the full-size tests, invisible on small grids.
    const int pipe = (K == 2 || p.stream3_pipe) ? 1 : 0;
    const int nw = stream3_waves(p, K, pipe);
    const int ns = stream3_slots(K, nw, pipe, p.stream3_slots);
#define LORA_S3(KK, WW, SS, PP)                                                                                       \
    if (K == KK && nw == WW && ns == SS && pipe == PP)                                                                \
        return p.tapset == TAPS3D_STAR                                                                                \
                   ? launch_stream3<TAPS3D_STAR, KK, WW, SS, PP != 0>(p, in, out, halo_src, parity, begin, end, s)    \
               : p.sep64_valid                                                                                         \
                   ? launch_stream3<TAPS3D_SEP, KK, WW, SS, PP != 0>(p, in, out, halo_src, parity, begin, end, s)     \
                   : launch_stream3<TAPS3D_BOX, KK, WW, SS, PP != 0>(p, in, out, halo_src, parity, begin, end, s);
    if (p.stream3_async && (nw == 8 || nw == 4)) {  // barrier-free form: 8 waves (one workgroup per CU) or 4 (two)
        const bool star = p.tapset == TAPS3D_STAR;
        if (K == 3 && nw == 8)
            return star ? launch_stream3<TAPS3D_STAR, 3, 8, 0, false>(p, in, out, halo_src, parity, begin, end, s)
                        : launch_stream3<TAPS3D_BOX, 3, 8, 0, false>(p, in, out, halo_src, parity, begin, end, s);
        if (K == 3)
            return star ? launch_stream3<TAPS3D_STAR, 3, 4, 0, false>(p, in, out, halo_src, parity, begin, end, s)
                        : launch_stream3<TAPS3D_BOX, 3, 4, 0, false>(p, in, out, halo_src, parity, begin, end, s);
        if (nw == 8)
            return star ? launch_stream3<TAPS3D_STAR, 2, 8, 0, false>(p, in, out, halo_src, parity, begin, end, s)
                        : launch_stream3<TAPS3D_BOX, 2, 8, 0, false>(p, in, out, halo_src, parity, begin, end, s);
        return star ? launch_stream3<TAPS3D_STAR, 2, 4, 0, false>(p, in, out, halo_src, parity, begin, end, s)
                    : launch_stream3<TAPS3D_BOX, 2, 4, 0, false>(p, in, out, halo_src, parity, begin, end, s);
    }
    LORA_S3(3, 8, 2, 0)
    LORA_S3(3, 4, 2, 0)
    LORA_S3(3, 6, 2, 1)
    LORA_S3(2, 8, 2, 1)
    LORA_S3(2, 4, 2, 1)
#undef LORA_S3
    return hipErrorInvalidValue;
}

const char *kernel_name_3d_stream(const Plan &) { return "stencil3d_planes_kernel"; }

}  // namespace lora
