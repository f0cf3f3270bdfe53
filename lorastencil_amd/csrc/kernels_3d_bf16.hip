// kernels_3d_bf16.hip -- 3D radius-1 sweeps on bf16 grids (BASELINE config 5: box3d1r 768^3 bf16).
// NEW capability: the reference is fp64-only (`#define DATA_TYPE double` is never used).
//
// Numerics (shared with the bf16 oracle, oracle_step_3d_bf16): values are STORED as bf16; one sweep converts the
// 27 neighbours to fp32 (exact), accumulates them with fp32 fused multiply-adds in the reference's tap order
// (dz, dy, dx; 3d/main.cu:33-68) against fp32 taps, and rounds ONCE to bf16 (round-to-nearest-even) on store.
// The same order and roundings make the result bit-identical to the oracle.
//
// Separable taps (TAPS3D_SEP): when the fp32 taps factor EXACTLY as w[dz][dy][dx] = a[dz] * b[dy] * c[dx]
// (weights.cpp separable_27; every box3d1r the reference's API can express does, because gpu_box_3d1r honours only
// params[0..2] = c, 3d/gpu_box.cu:143-226), the sweep is evaluated as the rank-1 product it is -- the 3D form of
// the paper's low-rank idea: T = x-pass (3 FMAs per staged element), U = y-pass of T, out = z-pass of U; 10.5
// instead of 27 multiply-adds per point.  The 27-tap form spends 0.275 ms of a 0.45 ms 768^3 sweep in the vector
// pipe with loads and stores removed (option `ablate`), so the tap count is what bounds it.  The order is part of
// the contract: T = fma(c2,x+,fma(c1,x0,c0*x-)), U and out likewise over y and z, all fp32, one rounding to bf16;
// oracle_step_3d_bf16_sep restates exactly that.
//
// Why not bf16 MFMA: the in-plane product (H X) V on v_mfma_f32_16x16x32_bf16 needs its
// first-stage result as a bf16 operand of the second stage -- an extra rounding of the intermediate that the
// "fp32 accumulate" contract does not have -- and at 4 algorithmic bytes per point the sweep is HBM-bound on the
// packed fp32 vector pipe anyway (27 FMAs per point = 36 % of the fp32 vector peak at the HBM roofline).
//
// Structure = kernels_3d.hip with 2-byte elements: a 256-thread workgroup owns a 16 x 256 column (lanes own 4
// adjacent columns, so a wave still moves 512 B-1 KiB contiguous row pieces), streams zc+2 input planes through a
// double-buffered LDS tile kept in bf16 (18 x 264 x 2 B: twice the bytes in flight per LDS byte of an fp32 tile),
// keeps three rotating sets of fp32 output-plane accumulators in registers and stores 8 bytes per lane per row.
// The innermost extent must be a multiple of 8 (16-byte row pieces).
#include <hip/hip_runtime.h>

#include "device_common.h"

namespace lora {

namespace {

typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Geometry for CPL columns per lane (4 or 8): a wave covers 64 x CPL columns, i.e. 512 B or 1 KiB row pieces.
template <int CPL>
struct Geo {
    static constexpr int kTileW = 64 * CPL;         // output columns per tile
    static constexpr int kLdsW = kTileW + 8;        // staged bf16 columns
    static constexpr int kChunksPerRow = kLdsW / 8; // 16-byte pieces per staged row
    static constexpr int kWinDwords = (CPL + 8) / 2; // window of a lane: elements 0 .. CPL+7 (own: 4 .. CPL+3)
};
constexpr int kTileW = Geo<4>::kTileW;   // the LDS-DMA ring kernel below is the 4-column form
constexpr int kLdsW = Geo<4>::kLdsW;
constexpr int kChunksPerRow = Geo<4>::kChunksPerRow;

struct Taps27f {
    float w[27];
};

template <int TAPSET>
__host__ __device__ constexpr bool tap_on3(int dz, int dy, int dx) {
    return TAPSET == TAPS3D_BOX ? true : (((dz != 1) + (dy != 1) + (dx != 1)) <= 1);
}

__device__ __forceinline__ float bf16_lo(unsigned pair) { return __builtin_bit_cast(float, pair << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pair) { return __builtin_bit_cast(float, pair & 0xffff0000u); }
// element e of a window held as bf16 pairs d[e / 2]
__device__ __forceinline__ float win_elem(const unsigned *d, int e) { return (e & 1) ? bf16_hi(d[e >> 1]) : bf16_lo(d[e >> 1]); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
    const u16 l = __builtin_bit_cast(u16, (__bf16) lo);
    const u16 h = __builtin_bit_cast(u16, (__bf16) hi);
    return (unsigned) l | ((unsigned) h << 16);
}

struct Args3Dh {
    const u16 *in;
    u16 *out;
    int h, m, n;
    int ld;
    long plane;
    int z_begin, z_end;
    int zc;
    int tiles_x, tiles_y;
    int ablate;  // timing-only diagnostics: 1 = no stores, 2 = no plane loads (results are then wrong)
    int dirichlet;  // fused kernel: level-1 cells outside the interior keep the source's halo value instead of 0
};

// acc + w * x as ONE unpacked v_fmac_f32 (inline asm: hipcc's SLP pass would otherwise re-pack two of them into a
// v_pk_fma_f32 and pay for the odd-aligned register pair with v_movs)
__device__ __forceinline__ float fmac_scalar(float w, float x, float acc) {
    asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "s"(w), "v"(x));
    return acc;
}

// One staged row `j` of plane phase PHASE applied to the rotating output-plane accumulators of a lane.
// pr[k] = window elements (3+k, 4+k) as an fp32 pair; NP column pairs per lane.  Called from fully unrolled loops,
// so j and every register index is a compile-time constant after inlining.
template <int TAPSET, int RY, int NP, int PHASE>
__device__ __forceinline__ void accumulate_row(f2 (&acc)[3][RY][NP], f2 (&u)[RY][NP], const f2 *pr, int j,
                                               const Taps27f &W) {
    if constexpr (TAPSET == TAPS3D_SEP) {
        // W.w[0..2] = c (x), [3..5] = b (y), [6..8] = a (z)
        // x-pass.  The outer taps read the window as the ALIGNED pairs (3,4), (5,6), .. -- each element converted once,
        // one packed op per pair; the middle tap straddles two pairs and is applied per element (a packed op would
        // need every element converted a second time into an odd-aligned pair: 2 conversions + 1 op instead of 2 ops)
        f2 t[NP];
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            t[c] = (f2){W.w[0], W.w[0]} * pr[2 * c];
            t[c].x = fmac_scalar(W.w[1], pr[2 * c].y, t[c].x);
            t[c].y = fmac_scalar(W.w[1], pr[2 * c + 2].x, t[c].y);
            t[c] = __builtin_elementwise_fma((f2){W.w[2], W.w[2]}, pr[2 * c + 2], t[c]);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int r = j - dy;
            if (r >= 0 && r < RY) {
                const f2 b2 = (f2){W.w[3 + dy], W.w[3 + dy]};
#pragma unroll
                for (int c = 0; c < NP; ++c) u[r][c] = dy == 0 ? b2 * t[c] : __builtin_elementwise_fma(b2, t[c], u[r][c]);
            }
        }
        if (j >= 2) {  // U row j-2 is complete: z-pass into the three output planes it touches
            const int r = j - 2;
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const int s = (PHASE - dz + 3) % 3;
                const f2 a2 = (f2){W.w[6 + dz], W.w[6 + dz]};
#pragma unroll
                for (int c = 0; c < NP; ++c)
                    acc[s][r][c] = dz == 0 ? a2 * u[r][c] : __builtin_elementwise_fma(a2, u[r][c], acc[s][r][c]);
            }
        }
    } else {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            const int s = (PHASE - dz + 3) % 3;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int dy = j - r;
                if (dy >= 0 && dy < 3) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if (tap_on3<TAPSET>(dz, dy, dx)) {
                            const float wt = W.w[dz * 9 + dy * 3 + dx];
                            const f2 wt2 = (f2){wt, wt};
#pragma unroll
                            for (int c = 0; c < NP; ++c)
                                acc[s][r][c] = __builtin_elementwise_fma(wt2, pr[2 * c + dx], acc[s][r][c]);
                        }
                    }
                }
            }
        }
    }
}

template <int TAPSET, int RY, int CPL>
__global__ __launch_bounds__(256, (CPL == 4 ? 4 : 3)) void stencil3d_bf16_kernel(const Args3Dh a, const Taps27f W) {
    using G = Geo<CPL>;
    constexpr int TY = 4 * RY;
    constexpr int LH = TY + 2;
    constexpr int NCHUNK = LH * G::kChunksPerRow;
    constexpr int NIT = (NCHUNK + 255) / 256;
    constexpr int ND = G::kWinDwords;
    __shared__ u32x4 tile[2][NCHUNK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;

    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int per_chunk = a.tiles_x * a.tiles_y;
    const int chunk = lin / per_chunk;
    const int rem = lin - chunk * per_chunk;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;
    const int i0 = ty * TY;
    const int j0 = tx * G::kTileW;
    const int zc = min(a.zc, a.z_end - k0);
    const int nplanes = zc + 2;

    long goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * 256;
        const int r = k / G::kChunksPerRow;
        const int c = k - r * G::kChunksPerRow;
        const int gr = min(i0 + 1 + r, a.m + 3);  // padded rows i0+1 .. i0+TY+2
        const int gc = min(j0 + 8 * c, a.n);      // padded columns j0 .. in 8-element pieces
        goff[it] = (long) gr * a.ld + gc;
    }
    u32x4 stage[NIT];
    auto load_plane = [&](int p) {
        const u16 *src = a.in + (long) min(k0 + p, a.h + 1) * a.plane;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (NCHUNK % 256 == 0 || tid + it * 256 < NCHUNK) stage[it] = *reinterpret_cast<const u32x4 *>(src + goff[it]);
        }
    };
    auto write_plane = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = tid + it * 256;
            if (NCHUNK % 256 == 0 || k < NCHUNK) tile[buf][k] = stage[it];
        }
    };

    // fp32 accumulators as column PAIRS (v_pk_fma_f32 applies one tap to two adjacent columns per instruction)
    f2 acc[3][RY][CPL / 2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int c = 0; c < CPL / 2; ++c) acc[s][r][c] = (f2){0.0f, 0.0f};

    const int col = j0 + CPL * lane;
    const bool col_ok = col < a.n;
    // window of a lane: tile columns CPL*lane .. CPL*lane+CPL+7 (own columns start at +4), 8-byte reads
    const int strip_off = (wv * RY) * G::kLdsW + CPL * lane;
    u16 *const out_col = a.out + (long) (i0 + wv * RY + 2) * a.ld + (col + 4);

    load_plane(0);
    write_plane(0);
    __syncthreads();

    auto consume = [&](int p, auto phase_tag) {
        constexpr int PHASE = decltype(phase_tag)::value;
        const bool more = p + 1 < nplanes;
        if (more && !(LORA_ABLATE(a) & 2)) load_plane(p + 1);
        const u16 *strip = reinterpret_cast<const u16 *>(&tile[p & 1][0]) + strip_off;
        f2 u[RY][CPL / 2];  // y-pass partial sums of this plane (separable form only)
#pragma unroll
        for (int j = 0; j < RY + 2; ++j) {
            // d[i] holds window elements (2i, 2i+1).  A tap dx of column pair c needs elements (3+2c+dx, 4+2c+dx):
            // pairs starting at 3 .. CPL+3, built as register pairs so that packed FMAs read them directly
            unsigned d[ND];
#pragma unroll
            for (int q = 0; q < ND / 2; ++q) {
                const u32x2 v = *reinterpret_cast<const u32x2 *>(strip + j * G::kLdsW + 4 * q);
                d[2 * q] = v.x;
                d[2 * q + 1] = v.y;
            }
            f2 pr[CPL + 1];  // pr[k] = elements (3+k, 4+k)
#pragma unroll
            for (int k = 0; k < CPL + 1; ++k) pr[k] = (f2){win_elem(d, 3 + k), win_elem(d, 4 + k)};
            accumulate_row<TAPSET, RY, CPL / 2, PHASE>(acc, u, pr, j, W);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < RY; ++r)
#pragma unroll
                for (int c = 0; c < CPL / 2; ++c) asm volatile("" : "+v"(acc[s][r][c]));

        {
            constexpr int s = (PHASE - 2 + 3) % 3;
            const int o = p - 2;
            if (o >= 0 && o < zc && col_ok && !(LORA_ABLATE(a) & 1)) {
                u16 *dst = out_col + (long) (k0 + o + 1) * a.plane;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (i0 + wv * RY + r < a.m) {
#pragma unroll
                        for (int h = 0; h < CPL / 4; ++h) {  // 8 bytes (4 columns) per store
                            u32x2 v;
                            // W.w[9]: 1 (exact) unless the plan asks for the matrix-pipe variant's contract
                            // (normalised factors, one scaling of the sum: kernels_3d_bf16_mfma.hip)
                            const float sc = TAPSET == TAPS3D_SEP ? W.w[9] : 1.0f;
                            v.x = pack_bf16(acc[s][r][2 * h].x * sc, acc[s][r][2 * h].y * sc);
                            v.y = pack_bf16(acc[s][r][2 * h + 1].x * sc, acc[s][r][2 * h + 1].y * sc);
                            *reinterpret_cast<u32x2 *>(dst + (long) r * a.ld + 4 * h) = v;
                        }
                    }
                }
            }
            if constexpr (TAPSET != TAPS3D_SEP) {  // the separable form assigns on its first z tap
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int c = 0; c < CPL / 2; ++c) acc[s][r][c] = (f2){0.0f, 0.0f};
            }
        }
        if (more) write_plane((p + 1) & 1);
        __syncthreads();
    };

    for (int p = 0; p < nplanes; p += 3) {
        consume(p, std::integral_constant<int, 0>{});
        if (p + 1 < nplanes) consume(p + 1, std::integral_constant<int, 1>{});
        if (p + 2 < nplanes) consume(p + 2, std::integral_constant<int, 2>{});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA ring variant.  The register-staged kernel above keeps ONE plane tile (9.5 KB) per workgroup in flight; with
// 5 workgroups per CU that is ~47 KB per CU against a loaded HBM latency of ~4 us, i.e. ~4 TB/s -- exactly where it
// sits (PMC), whatever the FMA count.  Here the planes travel global -> LDS directly (global_load_lds_dwordx4, no
// staging registers) into a 3-slot ring, TWO planes ahead of the one being consumed.  LDS-DMA completion is only
// ordered by the issuing wave's vmcnt, so the waits are hand-counted -- over the LOADS only: "plane p has landed" is
// s_waitcnt vmcnt(NIT), the pieces of plane p + 1 being the only younger loads (see consume()); a raw s_barrier (not
// __syncthreads, which would drain vmcnt(0)) then publishes every wave's part of the plane.
// ---------------------------------------------------------------------------------------------------------------
template <int TAPSET, int RY>
__global__ __launch_bounds__(256, 4) void stencil3d_bf16_ring_kernel(const Args3Dh a, const Taps27f W) {
    constexpr int TY = 4 * RY;
    constexpr int LH = TY + 2;
    constexpr int NCHUNK = LH * kChunksPerRow;
    constexpr int NIT = (NCHUNK + 255) / 256;
    constexpr int SLOT = NIT * 256;  // DMA writes whole 64-lane pieces: the tail of a slot is slack
    __shared__ u32x4 ring[3][SLOT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;

    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int per_chunk = a.tiles_x * a.tiles_y;
    const int chunk = lin / per_chunk;
    const int rem = lin - chunk * per_chunk;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;
    const int i0 = ty * TY;
    const int j0 = tx * kTileW;
    const int zc = min(a.zc, a.z_end - k0);
    const int nplanes = zc + 2;

    long goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = min(tid + it * 256, NCHUNK - 1);  // lanes past the tile re-read its last piece into the slack
        const int r = k / kChunksPerRow;
        const int c = k - r * kChunksPerRow;
        const int gr = min(i0 + 1 + r, a.m + 3);
        const int gc = min(j0 + 8 * c, a.n);
        goff[it] = (long) gr * a.ld + gc;
    }
    auto issue_plane = [&](int p, int slot) {
        const u16 *src = a.in + (long) min(k0 + p, a.h + 1) * a.plane;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            // LDS destination = wave-uniform base (M0) + lane * 16: this wave's 64 pieces of iteration `it`
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + goff[it]),
                                             (__attribute__((address_space(3))) void *) &ring[slot][it * 256 + (tid & ~63)],
                                             16, 0, 0);
        }
    };

    f2 acc[3][RY][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[s][r][c] = (f2){0.0f, 0.0f};

    const int col = j0 + 4 * lane;
    const int strip_off = (wv * RY) * kLdsW + 4 * lane;
    unsigned store_off[RY];  // byte offset inside an output plane, or out of range for lanes that must not store
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int row = i0 + wv * RY + r;
        store_off[r] = (col < a.n && row < a.m) ? (unsigned) (((long) (row + 2) * a.ld + (col + 4)) * 2) : 0xffffffffu;
    }
    const unsigned plane_bytes = (unsigned) (a.plane * 2);

    issue_plane(0, 0);
    issue_plane(1, 1);

    auto consume = [&](int p, auto phase_tag) {
        constexpr int PHASE = decltype(phase_tag)::value;  // = p mod 3 = ring slot of plane p
        // "plane p has landed": count LOADS only.  The loads younger than plane p's DMA are the NIT pieces of plane p + 1
        // (none for the last plane); loads complete in order among themselves, so with at most NIT operations outstanding
        // plane p is complete.  Counting the interleaved stores as well (vmcnt(2 RY + NIT), the first version of this
        // kernel) assumes a younger store never completes before an older LDS-DMA load: the same assumption gave a wrong
        // plane about once in 400 runs in kernels_3d_bf16_mfma.hip (tools/stress_bf16_mfma.py).
        if (p == nplanes - 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (p + 2 < nplanes) issue_plane(p + 2, (PHASE + 2) % 3);  // slot of plane p-1: every wave is done with it
        const u16 *strip = reinterpret_cast<const u16 *>(&ring[PHASE][0]) + strip_off;
        f2 u[RY][2];
#pragma unroll
        for (int j = 0; j < RY + 2; ++j) {
            unsigned d[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x2 v = *reinterpret_cast<const u32x2 *>(strip + j * kLdsW + 4 * q);
                d[2 * q] = v.x;
                d[2 * q + 1] = v.y;
            }
            f2 pr[5];
            pr[0] = (f2){bf16_hi(d[1]), bf16_lo(d[2])};
            pr[1] = (f2){bf16_lo(d[2]), bf16_hi(d[2])};
            pr[2] = (f2){bf16_hi(d[2]), bf16_lo(d[3])};
            pr[3] = (f2){bf16_lo(d[3]), bf16_hi(d[3])};
            pr[4] = (f2){bf16_hi(d[3]), bf16_lo(d[4])};
            accumulate_row<TAPSET, RY, 2, PHASE>(acc, u, pr, j, W);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < RY; ++r) asm volatile("" : "+v"(acc[s][r][0]), "+v"(acc[s][r][1]));
        {
            constexpr int s = (PHASE - 2 + 3) % 3;
            const int o = p - 2;
            const bool live = o >= 0 && o < zc;
            const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
                a.out + (long) min(max(k0 + o + 1, 0), a.h) * a.plane, 0, live ? plane_bytes : 0u, 0x00020000);
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                u32x2 v;
                const float sc = TAPSET == TAPS3D_SEP ? W.w[9] : 1.0f;
                v.x = pack_bf16(acc[s][r][0].x * sc, acc[s][r][0].y * sc);
                v.y = pack_bf16(acc[s][r][1].x * sc, acc[s][r][1].y * sc);
                __builtin_amdgcn_raw_buffer_store_b64(v, dst, store_off[r], 0, 0);
            }
            if constexpr (TAPSET != TAPS3D_SEP) {
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[s][r][c] = (f2){0.0f, 0.0f};
            }
        }
    };

    for (int p = 0; p < nplanes; p += 3) {
        consume(p, std::integral_constant<int, 0>{});
        if (p + 1 < nplanes) consume(p + 1, std::integral_constant<int, 1>{});
        if (p + 2 < nplanes) consume(p + 2, std::integral_constant<int, 2>{});
    }
}

template <int TAPSET, int CPL>
hipError_t launch_bf16(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s) {
    constexpr int RY = 4, TY = 4 * RY;
    constexpr int kTileW = Geo<CPL>::kTileW;
    Args3Dh a;
    a.in = static_cast<const u16 *>(in);
    a.out = static_cast<u16 *>(out);
    a.h = p.dims[0];
    a.m = p.dims[1];
    a.n = p.dims[2];
    a.ld = a.n + 8;
    a.plane = (long) (a.m + 4) * (a.n + 8);
    a.z_begin = begin;
    a.z_end = end;
    a.zc = p.z_chunk < 1 ? 1 : p.z_chunk;
    a.tiles_x = (a.n + kTileW - 1) / kTileW;
    a.tiles_y = (a.m + TY - 1) / TY;
    a.ablate = p.ablate;
    a.dirichlet = 0;
    const long chunks = ((long) end - begin + a.zc - 1) / a.zc;
    const long nblocks = chunks * a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    Taps27f w;
    for (int k = 0; k < 27; ++k) w.w[k] = (float) p.w[k];
    if (TAPSET == TAPS3D_SEP) {
        for (int k = 0; k < 9; ++k) w.w[k] = p.sep[k];
        w.w[9] = 1.0f;
        if (p.variant == LORA_VARIANT_MFMA && p.mfma3_valid) {
            // single sweeps of a plan that runs the matrix-pipe variant (odd tails) follow ITS contract: the sum with
            // the normalised factors (exact in the regime that contract is stated for), scaled once
            for (int k = 0; k < 9; ++k) w.w[k] = p.mfma3_abc[k];
            w.w[9] = p.mfma3_scale;
        }
    }
    if (a.plane * 2 >= (1L << 32)) return hipErrorInvalidValue;  // per-plane buffer descriptors: 32-bit offsets
    if (CPL == 4 && p.lds_dma)
        hipLaunchKernelGGL((stencil3d_bf16_ring_kernel<TAPSET, RY>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    else
        hipLaunchKernelGGL((stencil3d_bf16_kernel<TAPSET, RY, CPL>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// TWO applications per launch (temporal fusion), the bf16 counterpart of kernels_3d_fused.hip.  Time level 1 is
// rounded to bf16 exactly as a single sweep would store it and lives in LDS tile B in the input's format, so both
// levels run the same window code and the result is bit-identical to two single bf16 sweeps (either summation
// form).  Level-1 cells outside the interior are 0 (the reference driver's "buffer 1" halo).
//
// Geometry (256 threads; a wave = 2 row groups x 32 lanes, a lane owns 4 adjacent columns x RY rows, 8 strips):
//   output tile    8 RY - 2 rows x 120 columns (lanes 1..30 of a row group), padded origin (2 + 30 ty, 4 + 120 tx)
//   level-1 tile   8 RY rows     x 128 columns, starts 1 row / 4 columns earlier (padded column = 0 mod 8)
//   input window   8 RY + 2 rows x 144 columns, starts 2 rows / 12 columns earlier: whole 16-byte pieces
// A lane's level-1 and level-2 columns coincide (padded 120 tx + 4 cl), so both levels read the 12-element window
// at tile column 4 cl + 4 of A resp. B.  Plane pipeline and accumulator rotation: see kernels_3d_fused.hip.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFusedLanes = 32;
constexpr int kFusedOutW = 4 * (kFusedLanes - 2);  // 120
constexpr int kFusedLdsW = 4 * kFusedLanes + 16;   // 144 staged columns = LDS row stride of A and B
constexpr int kFusedChunks = kFusedLdsW / 8;       // 18 pieces of 16 bytes per row

// PIPE: both tiles double-buffered (39 KB) and level 2 runs one plane behind level 1 -- iteration p sweeps input
// plane p from A[p & 1] AND level-1 plane p-2 from B[(p-1) & 1] back to back, then publishes level-1 plane p-1 in
// B[p & 1] and refills A[(p+1) & 1]: ONE barrier per plane instead of two, at the price of one more iteration.
template <int TAPSET, int RY, bool PIPE, bool DIRICHLET>
__global__ __launch_bounds__(256, 3) void stencil3d_bf16_fused2_kernel(const Args3Dh a, const Taps27f W) {
    constexpr int MH = 8 * RY;
    constexpr int OH = MH - 2;
    constexpr int IH = MH + 2;
    constexpr int NCHUNK = IH * kFusedChunks;
    constexpr int NIT = (NCHUNK + 255) / 256;
    __shared__ u32x4 A[PIPE ? 2 : 1][NCHUNK];
    __shared__ u32x4 B[PIPE ? 2 : 1][NCHUNK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int sid = wv * 2 + lane / kFusedLanes;
    const int cl = lane % kFusedLanes;

    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int per_chunk = a.tiles_x * a.tiles_y;
    const int chunk = lin / per_chunk;
    const int rem = lin - chunk * per_chunk;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;
    const int zc = min(a.zc, a.z_end - k0);
    const int I = ty * OH;          // first output row (interior)
    const int J = tx * kFusedOutW;  // first output column (interior), multiple of 8

    int goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * 256;
        const int r = k / kFusedChunks;
        const int c = k - r * kFusedChunks;
        const int gr = min(I + r, a.m + 3);                  // padded rows I .. I + IH - 1
        const int gc = min(max(J - 8 + 8 * c, 0), a.n);      // padded columns J - 8 .. in 8-element pieces
        goff[it] = gr * a.ld + gc;
    }
    u32x4 stage[NIT];
    auto load_plane = [&](int p) {
        const u16 *src = a.in + (long) min(max(k0 - 1 + p, 0), a.h + 1) * a.plane;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (NCHUNK % 256 == 0 || tid + it * 256 < NCHUNK) stage[it] = *reinterpret_cast<const u32x4 *>(src + goff[it]);
        }
    };
    auto write_plane = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = tid + it * 256;
            if (NCHUNK % 256 == 0 || k < NCHUNK) A[buf][k] = stage[it];
        }
    };

    f2 acc1[3][RY][2], acc2[3][RY][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc1[s][r][c] = acc2[s][r][c] = (f2){0.0f, 0.0f};

    // this lane's 4 columns at BOTH levels: interior J - 4 + 4 cl .. (tile column 4 cl + 8)
    const int col = J - 4 + 4 * cl;
    const bool col_in = col >= 0 && col < a.n;                       // n is a multiple of 8
    const bool col_out = cl >= 1 && cl <= kFusedLanes - 2 && col < a.n;
    const int row1 = I - 1 + sid * RY;  // first level-1 row (interior) of the strip
    const int rowo = I + sid * RY;      // first output row
    const int strip_off = (sid * RY) * kFusedLdsW + 4 * cl + 4;  // window = tile columns 4 cl + 4 .. 4 cl + 15
    u16 *const out_col = a.out + (long) (rowo + 2) * a.ld + (col + 4);

    load_plane(0);
    write_plane(0);
    __syncthreads();

    auto sweep_tile = [&](const u32x4 *tile, f2 (&acc)[3][RY][2], auto phase_tag) {
        constexpr int PHASE = decltype(phase_tag)::value;
        const u16 *strip = reinterpret_cast<const u16 *>(tile) + strip_off;
        f2 u[RY][2];
#pragma unroll
        for (int j = 0; j < RY + 2; ++j) {
            unsigned d[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x2 v = *reinterpret_cast<const u32x2 *>(strip + j * kFusedLdsW + 4 * q);
                d[2 * q] = v.x;
                d[2 * q + 1] = v.y;
            }
            f2 pr[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) pr[k] = (f2){win_elem(d, 3 + k), win_elem(d, 4 + k)};
            accumulate_row<TAPSET, RY, 2, PHASE>(acc, u, pr, j, W);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < RY; ++r) asm volatile("" : "+v"(acc[s][r][0]), "+v"(acc[s][r][1]));
    };

    // level-1 plane of interior index z1, complete in slot `s` of acc1: round to bf16, 0 outside the interior
    auto publish = [&](u32x4 *tile, int z1, auto slot_tag) {
        constexpr int s = decltype(slot_tag)::value;
        const bool z_in = z1 >= 0 && z1 < a.h;
        u16 *dstB = reinterpret_cast<u16 *>(tile) + strip_off + 4;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const bool in = z_in && col_in && row1 + r >= 0 && row1 + r < a.m;
            u32x2 v;
            v.x = in ? pack_bf16(acc1[s][r][0].x, acc1[s][r][0].y) : 0u;
            v.y = in ? pack_bf16(acc1[s][r][1].x, acc1[s][r][1].y) : 0u;
            if (DIRICHLET && !in) {  // fixed boundary: halo cells keep the source's value at every level (template
                                     // flag: as a run-time one it costs the reference path 9 VGPRs = one workgroup per CU)
                const int pz = z1 + 1, pr = row1 + r + 2, pc = col + 4;
                if (pz >= 0 && pz <= a.h + 1 && pr >= 0 && pr <= a.m + 3 && pc >= 0 && pc + 3 <= a.n + 7)
                    v = *reinterpret_cast<const u32x2 *>(a.in + (long) pz * a.plane + (long) pr * a.ld + pc);
            }
            *reinterpret_cast<u32x2 *>(dstB + r * kFusedLdsW) = v;
            if constexpr (TAPSET != TAPS3D_SEP) acc1[s][r][0] = acc1[s][r][1] = (f2){0.0f, 0.0f};
        }
    };
    // output plane o of the chunk, complete in slot `s` of acc2
    auto store_plane = [&](int o, auto slot_tag) {
        constexpr int s = decltype(slot_tag)::value;
        if (o >= 0 && o < zc && col_out && !(LORA_ABLATE(a) & 1)) {
            u16 *dst = out_col + (long) (k0 + o + 1) * a.plane;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                if (sid * RY + r < OH && rowo + r < a.m) {
                    u32x2 v;
                    v.x = pack_bf16(acc2[s][r][0].x, acc2[s][r][0].y);
                    v.y = pack_bf16(acc2[s][r][1].x, acc2[s][r][1].y);
                    *reinterpret_cast<u32x2 *>(dst + (long) r * a.ld) = v;
                }
            }
        }
        if constexpr (TAPSET != TAPS3D_SEP) {
#pragma unroll
            for (int r = 0; r < RY; ++r) acc2[s][r][0] = acc2[s][r][1] = (f2){0.0f, 0.0f};
        }
    };

    const int nin = zc + 4;                   // input planes of the chunk
    const int niter = PIPE ? nin + 1 : nin;
    auto consume = [&](int p, auto phase_tag) {
        constexpr int PH = decltype(phase_tag)::value;  // p mod 3
        const bool more = p + 1 < nin;
        if (more && !(LORA_ABLATE(a) & 2)) load_plane(p + 1);
        if constexpr (PIPE) {
            if (p < nin) sweep_tile(A[p & 1], acc1, std::integral_constant<int, PH>{});
            // level-1 plane p-2 (phase (p-2) mod 3), published during the previous iteration
            if (p >= 1) sweep_tile(B[(p - 1) & 1], acc2, std::integral_constant<int, (PH + 1) % 3>{});
            if (p < nin) publish(B[p & 1], k0 - 3 + p, std::integral_constant<int, (PH + 1) % 3>{});
            if (more) write_plane((p + 1) & 1);
            store_plane(p - 5, std::integral_constant<int, (PH + 2) % 3>{});  // slot ((p-2) - 2) mod 3
            __syncthreads();
        } else {
            sweep_tile(A[0], acc1, std::integral_constant<int, PH>{});
            publish(B[0], k0 - 3 + p, std::integral_constant<int, (PH + 1) % 3>{});
            __syncthreads();
            sweep_tile(B[0], acc2, std::integral_constant<int, (PH + 2) % 3>{});
            // refill A first: its wait for the prefetched plane then does not include this iteration's stores
            if (more) write_plane(0);
            store_plane(p - 4, std::integral_constant<int, PH>{});
            __syncthreads();
        }
    };

    for (int p = 0; p < niter; p += 3) {
        consume(p, std::integral_constant<int, 0>{});
        if (p + 1 < niter) consume(p + 1, std::integral_constant<int, 1>{});
        if (p + 2 < niter) consume(p + 2, std::integral_constant<int, 2>{});
    }
}

template <int TAPSET>
hipError_t launch_bf16_fused2(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s) {
    constexpr int RY = 4, OH = 8 * RY - 2;
    Args3Dh a;
    a.in = static_cast<const u16 *>(in);
    a.out = static_cast<u16 *>(out);
    a.h = p.dims[0];
    a.m = p.dims[1];
    a.n = p.dims[2];
    a.ld = a.n + 8;
    a.plane = (long) (a.m + 4) * (a.n + 8);
    if (a.plane >= (1L << 31)) return hipErrorInvalidValue;  // 32-bit in-plane offsets
    a.z_begin = begin;
    a.z_end = end;
    a.tiles_x = (a.n + kFusedOutW - 1) / kFusedOutW;
    a.tiles_y = (a.m + OH - 1) / OH;
    a.ablate = p.ablate;
    a.dirichlet = p.boundary == LORA_BC_DIRICHLET;
    int zc = p.fused_z_chunk;
    if (zc <= 0) {  // every chunk re-reads (and re-computes) 4 planes: long chunks while ~3 rounds of 3 per CU remain
        zc = 32;  // 768^3: 24-32 planes 1600 GStencils/s, 16 -> 1540, 48-64 -> 1490-1560
        const long per_plane = (long) a.tiles_x * a.tiles_y;
        while (zc > 8 && per_plane * ((end - begin + zc - 1) / zc) < 3 * 1024) zc /= 2;  // 1024 = 4 per CU
    }
    a.zc = zc;
    const long chunks = ((long) end - begin + a.zc - 1) / a.zc;
    const long nblocks = chunks * a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    Taps27f w;
    for (int k = 0; k < 27; ++k) w.w[k] = (float) p.w[k];
    if (TAPSET == TAPS3D_SEP)
        for (int k = 0; k < 9; ++k) w.w[k] = p.sep[k];
    if (a.dirichlet)
        hipLaunchKernelGGL((stencil3d_bf16_fused2_kernel<TAPSET, RY, false, true>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    else if (p.fused_pipeline)
        hipLaunchKernelGGL((stencil3d_bf16_fused2_kernel<TAPSET, RY, true, false>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    else
        hipLaunchKernelGGL((stencil3d_bf16_fused2_kernel<TAPSET, RY, false, false>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_3d_bf16(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s) {
    const bool wide = p.cols_per_lane == 8 && !p.lds_dma;
    if (p.tapset == TAPS3D_SEP)
        return wide ? launch_bf16<TAPS3D_SEP, 8>(p, in, out, begin, end, s) : launch_bf16<TAPS3D_SEP, 4>(p, in, out, begin, end, s);
    if (p.tapset == TAPS3D_STAR)
        return wide ? launch_bf16<TAPS3D_STAR, 8>(p, in, out, begin, end, s) : launch_bf16<TAPS3D_STAR, 4>(p, in, out, begin, end, s);
    return wide ? launch_bf16<TAPS3D_BOX, 8>(p, in, out, begin, end, s) : launch_bf16<TAPS3D_BOX, 4>(p, in, out, begin, end, s);
}

const char *kernel_name_3d_bf16(const Plan &) { return "stencil3d_bf16_kernel"; }

hipError_t launch_3d_bf16_fused2(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    if (p.tapset == TAPS3D_SEP) return launch_bf16_fused2<TAPS3D_SEP>(p, in, out, begin, end, s);
    if (p.tapset == TAPS3D_STAR) return launch_bf16_fused2<TAPS3D_STAR>(p, in, out, begin, end, s);
    return launch_bf16_fused2<TAPS3D_BOX>(p, in, out, begin, end, s);
}

const char *kernel_name_3d_bf16_fused2(const Plan &) { return "stencil3d_bf16_fused2_kernel"; }

}  // namespace lora
