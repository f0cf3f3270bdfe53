// kernels_3d_bf16.hip -- 3D radius-1 sweeps on bf16 grids (BASELINE config 5: box3d1r 768^3 bf16).
// NEW capability: the reference is fp64-only (`#define DATA_TYPE double` is never used).
//
// Numerics (shared with the bf16 oracle, oracle_step_3d_bf16): values are STORED as bf16; one sweep converts the
// 27 neighbours to fp32 (exact), accumulates them with fp32 fused multiply-adds in the reference's tap order
// (dz, dy, dx; 3d/main.cu:33-68) against fp32 taps, and rounds ONCE to bf16 (round-to-nearest-even) on store.
// The same order and roundings make the result bit-identical to the oracle.
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

#include "engine.h"

namespace lora {

namespace {

typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTileW = 256;              // output columns per tile: 64 lanes x 4
constexpr int kLdsW = kTileW + 8;        // staged bf16 columns
constexpr int kChunksPerRow = kLdsW / 8; // 16-byte pieces per staged row

struct Taps27f {
    float w[27];
};

template <int TAPSET>
__host__ __device__ constexpr bool tap_on3(int dz, int dy, int dx) {
    return TAPSET == TAPS3D_BOX ? true : (((dz != 1) + (dy != 1) + (dx != 1)) <= 1);
}

__device__ __forceinline__ int xcd_contiguous(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, slot = b >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

__device__ __forceinline__ float bf16_lo(unsigned pair) { return __builtin_bit_cast(float, pair << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pair) { return __builtin_bit_cast(float, pair & 0xffff0000u); }
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
};

template <int TAPSET, int RY>
__global__ __launch_bounds__(256, 4) void stencil3d_bf16_kernel(const Args3Dh a, const Taps27f W) {
    constexpr int TY = 4 * RY;
    constexpr int LH = TY + 2;
    constexpr int NCHUNK = LH * kChunksPerRow;
    constexpr int NIT = (NCHUNK + 255) / 256;
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
    const int j0 = tx * kTileW;
    const int zc = min(a.zc, a.z_end - k0);
    const int nplanes = zc + 2;

    long goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * 256;
        const int r = k / kChunksPerRow;
        const int c = k - r * kChunksPerRow;
        const int gr = min(i0 + 1 + r, a.m + 3);  // padded rows i0+1 .. i0+TY+2
        const int gc = min(j0 + 8 * c, a.n);      // padded columns j0 .. j0+263 in 8-element pieces
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

    float acc[3][RY][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[s][r][c] = 0.0f;

    const int col = j0 + 4 * lane;
    const bool col_ok = col < a.n;
    // window of a lane: tile columns 4*lane .. 4*lane+11 (own columns are 4*lane+4 .. +7), three 8-byte reads
    const int strip_off = (wv * RY) * kLdsW + 4 * lane;
    u16 *const out_col = a.out + (long) (i0 + wv * RY + 2) * a.ld + (col + 4);

    load_plane(0);
    write_plane(0);
    __syncthreads();

    auto consume = [&](int p, auto phase_tag) {
        constexpr int PHASE = decltype(phase_tag)::value;
        const bool more = p + 1 < nplanes;
        if (more) load_plane(p + 1);
        const u16 *strip = reinterpret_cast<const u16 *>(&tile[p & 1][0]) + strip_off;
#pragma unroll
        for (int j = 0; j < RY + 2; ++j) {
            float win[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x2 v = *reinterpret_cast<const u32x2 *>(strip + j * kLdsW + 4 * q);
                win[4 * q + 0] = bf16_lo(v.x);
                win[4 * q + 1] = bf16_hi(v.x);
                win[4 * q + 2] = bf16_lo(v.y);
                win[4 * q + 3] = bf16_hi(v.y);
            }
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
#pragma unroll
                                for (int c = 0; c < 4; ++c) acc[s][r][c] = fmaf(wt, win[3 + c + dx], acc[s][r][c]);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < RY; ++r)
                asm volatile("" : "+v"(acc[s][r][0]), "+v"(acc[s][r][1]), "+v"(acc[s][r][2]), "+v"(acc[s][r][3]));

        {
            constexpr int s = (PHASE - 2 + 3) % 3;
            const int o = p - 2;
            if (o >= 0 && o < zc && col_ok) {
                u16 *dst = out_col + (long) (k0 + o + 1) * a.plane;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (i0 + wv * RY + r < a.m) {
                        u32x2 v;
                        v.x = pack_bf16(acc[s][r][0], acc[s][r][1]);
                        v.y = pack_bf16(acc[s][r][2], acc[s][r][3]);
                        *reinterpret_cast<u32x2 *>(dst + (long) r * a.ld) = v;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RY; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[s][r][c] = 0.0f;
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

template <int TAPSET>
hipError_t launch_bf16(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s) {
    constexpr int RY = 4, TY = 4 * RY;
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
    const long chunks = ((long) end - begin + a.zc - 1) / a.zc;
    const long nblocks = chunks * a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    Taps27f w;
    for (int k = 0; k < 27; ++k) w.w[k] = (float) p.w[k];
    hipLaunchKernelGGL((stencil3d_bf16_kernel<TAPSET, RY>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_3d_bf16(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s) {
    if (p.tapset == TAPS3D_STAR) return launch_bf16<TAPS3D_STAR>(p, in, out, begin, end, s);
    return launch_bf16<TAPS3D_BOX>(p, in, out, begin, end, s);
}

const char *kernel_name_3d_bf16(const Plan &) { return "stencil3d_bf16_kernel"; }

}  // namespace lora
