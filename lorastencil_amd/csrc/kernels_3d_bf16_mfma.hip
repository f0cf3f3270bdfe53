// kernels_3d_bf16_mfma.hip -- box3d1r on bf16 grids, TWO applications per launch, in-plane passes on the matrix pipe
// (v_mfma_f32_16x16x32_bf16).  BASELINE config 5 names "CDNA4 bf16 MFMA"; the reference's own kernel is the fp64
// tensor-core form of the same idea: in-plane product (H X) V with the first product fed to the second from
// registers, 3d/gpu_box.cu:43-86, plane ring :105-140.  LORA_VARIANT_MFMA of a bf16 plan selects it.
//
// Applies to separable taps w = scale * a'(z) (x) b'(y) (x) c'(x) whose normalised factors a', b', c' are exact in
// bf16 (weights.cpp mfma_factors_27: the reference's box3d1r, (1 1 1)(1 1 1)(1 2 1), and every multiple of it, e.g.
// the normalised taps).  Numeric contract -- restated by oracle_step_3d_bf16_mfma:
//     S   = sum of the 27 products a'[dz] b'[dy] c'[dx] x, accumulated in fp32 by the matrix instruction (EXACT while
//           the addends of a point span fewer than ~14 binary orders of magnitude -- the usual case; the hi + lo split
//           below keeps 16 bits of the intermediate otherwise, i.e. an error 2^-9 of a bf16 ulp)
//     out = RNE_bf16( fl32( scale * S ) )
// i.e. the correctly rounded 27-point sum scaled once: at least as accurate as the vector kernel's 9-deep fp32 FMA
// chain, but a different fp32 rounding sequence, hence a contract of its own (bit-exact against its oracle on data in
// the exact regime, within one bf16 ulp otherwise).
//
// Shape of the computation (one WAVE owns a 28 x 60 output tile and walks z; no workgroup barriers):
//   * input plane tile 32 rows x 64 columns of bf16 (4 KB), fetched global -> LDS by 4 x global_load_lds_dwordx4 two
//     planes ahead into a 3-slot ring (hand-counted vmcnt over the LOADS only, as in kernels_2d_stream.hip);
//   * x-pass  P = X V      : A = 16 rows x 32 columns of the tile straight from LDS (ONE ds_read_b128 per lane IS the A
//                            fragment), B = the banded 3-tap matrix of c' (a constant fragment); 8 MFMAs per plane
//   * z-pass  T = sum a' P : on the fp32 accumulators of the last three planes (rotating register sets) -- so each
//                            plane goes through the matrix pipe once, not three times
//   * y-pass  U^T = T^T H^T: the x-pass accumulator layout (lane = column, 4 rows per register set) IS the A-fragment
//                            layout of the TRANSPOSED product once the band of b' is permuted to the register order
//                            (the CDNA4 form of the reference's V-row permutation, 2d/gpu.cu:506-519) -- the fp32
//                            values are split hi + lo into two bf16 fragments in place, no LDS round trip; the
//                            transposed result puts 4 consecutive COLUMNS of a row into one lane: 8-byte stores
//   * level 1 is rounded to bf16 as a single sweep would store it (cells outside the interior = 0) and
//     written to a one-plane LDS tile in the input's format; level 2 runs the same three passes on it and stores.
// Matrix-pipe work: 48 MFMAs (768 cycles) per plane and wave for 2 x 1680 point-applications.
#include <hip/hip_runtime.h>

#include "device_common.h"

namespace lora {

namespace {

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTR = 32, kTC = 64;        // input tile (rows, columns)
constexpr int kOR = kTR - 4, kOC = kTC - 4;  // output tile 28 x 60
constexpr int kPitch = kTC;              // LDS row pitch in elements (128 B)
constexpr int kSlot = kTR * kPitch;      // elements per plane tile (4 KB)
constexpr int kDepth = 2;                // input planes in flight
constexpr int kRing = kDepth + 1;        // ring slots (= the unroll factor of the plane loop)
constexpr int kWaveLds = (kRing + 1) * kSlot;  // ring + the level-1 tile, in elements
constexpr int kLoadsPerStep = 4;        // (and 8 stores per step)

struct ArgsMfma3 {
    const u16 *in;
    u16 *out;
    int h, m, n, ld;
    long plane;
    int z_begin, z_end, zc;
    int tiles_x, tiles_y;
    float scale;
    float a[3], b[3], c[3];  // normalised factors (exact in bf16)
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    // a vector conversion: ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN), no shifts / ors
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf16_round(float x) {  // fl32 of the bf16 nearest to x
    return (float) (__bf16) x;
}

// SPLIT: the fp32 intermediate goes to the second product as hi + lo bf16 halves (the contract above); false = one
// bf16 rounding of it (an experiment: what the split costs)
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void stencil3d_bf16_mfma2_kernel(const ArgsMfma3 a) {
    __shared__ __attribute__((aligned(16))) u16 lds[4 * kWaveLds + 64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u16 *const ring = lds + wv * kWaveLds;
    u16 *const tile1 = ring + kRing * kSlot;  // level-1 plane, origin shifted by (1, 1) against the input tile

    // wave -> (chunk, tile): four x-adjacent tiles per workgroup
    const long per_chunk = (long) a.tiles_x * a.tiles_y;
    const int nchunks = (a.z_end - a.z_begin + a.zc - 1) / a.zc;
    const long total = per_chunk * nchunks;
    const long wlin0 = (long) xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wv;
    const bool spare = wlin0 >= total;
    const long wlin = spare ? total - 1 : wlin0;
    const int chunk = (int) (wlin / per_chunk);
    const int rem = (int) (wlin - (long) chunk * per_chunk);
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;
    const int k_end = spare ? k0 : min(k0 + a.zc, a.z_end);
    const int i0 = ty * kOR, j0 = tx * kOC;  // first output row / column (interior)

    // ---- constant B fragments.  lane (n = lane % 16, kb = lane / 16) holds B[8 kb + jj][n], jj = 0..7 ----
    const int n16 = lane & 15, kb = lane >> 4;
    bf16x8 vx, vx16;  // x-pass: B[k][n] = c'[k - n] (vx16: c'[k - n - 16]), k - n in 0..2: output column t <- input
                      // columns t, t+1, t+2.  Column blocks 0..2 read the window starting at their first column;
                      // block 3 re-uses block 2's window with the band 16 further down -- no read runs past a row
    bf16x8 hy[2];  // y-pass (transposed product, K in accumulator-register order): B[k][n] = b'[rho(k) - u],
                   // u = 16 ib + n, rho(8 kb + jj) = jj < 4 ? 4 kb + jj : 16 + 4 kb + jj - 4
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int dk = 8 * kb + jj - n16;
        vx[jj] = (__bf16) (dk == 0 ? a.c[0] : (dk == 1 ? a.c[1] : (dk == 2 ? a.c[2] : 0.0f)));
        vx16[jj] = (__bf16) (dk == 16 ? a.c[0] : (dk == 17 ? a.c[1] : (dk == 18 ? a.c[2] : 0.0f)));
        const int rho = jj < 4 ? 4 * kb + jj : 16 + 4 * kb + (jj - 4);
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int dr = rho - (16 * ib + n16);
            hy[ib][jj] = (__bf16) (dr == 0 ? a.b[0] : (dr == 1 ? a.b[1] : (dr == 2 ? a.b[2] : 0.0f)));
        }
    }

    // ---- loads: plane step s = interior plane k0 - 2 + s; 4 instructions of 8 rows x 128 B ----
    const int lrow = lane >> 3, lq = lane & 7;
    // padded column of this lane's 16-byte piece.  Pieces may run past the end of a row into the next one (those
    // columns only feed outputs beyond the grid); only the very end of the ARRAY is clamped (the last pad row of the
    // last pad plane: read by nothing that is stored)
    const int gcol = j0 + 2 + 8 * lq;
    const long last_piece = (long) (a.h + 2) * a.plane - 8;
    auto issue = [&](int s, int slot) {
        const int pz = min(max(k0 - 1 + s, 0), a.h + 1);  // padded plane
        const long pbase = (long) pz * a.plane + gcol;
#pragma unroll
        for (int q = 0; q < kLoadsPerStep; ++q) {
            const int pr = min(i0 + 8 * q + lrow, a.m + 3);  // padded row (clamped)
            const long off = min(pbase + (long) pr * a.ld, last_piece);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (a.in + off),
                                             (__attribute__((address_space(3))) void *) (ring + slot * kSlot + q * 8 * kPitch),
                                             16, 0, 0);
        }
    };

    // ---- stores: block (j, ib): lane (n16, g) holds output row u = 16 ib + n16, columns t = 16 j + 4 g .. + 3 ----
    const int g4 = lane >> 4;
    unsigned store_off[4][2];
    bool l1_row_in[2];
    unsigned l1_col_in[4];  // 4 bits: columns t .. t + 3 of block j inside the interior at level 1
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
        const int u = 16 * ib + n16;
        l1_row_in[ib] = (i0 - 1 + u) >= 0 && (i0 - 1 + u) < a.m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = 16 * j + 4 * g4;
            const bool ok = u < kOR && t < kOC && (i0 + u) < a.m && (j0 + t) < a.n;
            store_off[j][ib] = ok ? (unsigned) ((((long) (i0 + u + 2)) * a.ld + (j0 + t + 4)) * 2) : 0x80000000u;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = j0 - 1 + 16 * j + 4 * g4 + i;
            bits |= (col >= 0 && col < a.n) ? (1u << i) : 0u;
        }
        l1_col_in[j] = bits;
    }
    const unsigned plane_bytes = (unsigned) (a.plane * 2);

    // rotating fp32 x-pass results of the last three planes, per level: [slot][row block][column block]
    f32x4 p1[3][2][4], p2[3][2][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) p1[s][rb][j] = p2[s][rb][j] = (f32x4){0, 0, 0, 0};
    {  // the level-1 tile starts as zeros (its first use reads what step 0 wrote; this keeps the slack finite)
        u32x4 z = {0, 0, 0, 0};
        for (int k = lane; k < kSlot / 8; k += 64) reinterpret_cast<u32x4 *>(tile1)[k] = z;
    }

    // vector-memory stream: planes 0 .. kDepth - 1 up front, then per step { 8 stores, the loads of plane s + kDepth }
    issue(0, 0);
#pragma unroll
    for (int k = 1; k < kDepth; ++k) issue(k, k);

    // x-pass of one plane tile: 6 ds_read_b128 + 8 MFMAs -> p[rb][j]
    auto xpass = [&](const u16 *tile, f32x4 (&p)[2][4]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(tile + (16 * rb + n16) * kPitch + 16 * j + 8 * kb);
                p[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, vx, (f32x4){0, 0, 0, 0}, 0, 0, 0);
                if (j == 2) p[rb][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, vx16, (f32x4){0, 0, 0, 0}, 0, 0, 0);
            }
    };
    // z-pass + y-pass for column block j: -> scale * U[u][t] for the two row blocks (4 values each)
    auto zy = [&](const f32x4 (&pm)[2][4], const f32x4 (&pc)[2][4], const f32x4 (&pn)[2][4], int j, f32x4 (&u)[2]) {
        // fragment dword d holds elements (2 d, 2 d + 1) = T rows (4 rb + i) of this lane's column: built pair-wise so
        // that one v_cvt_pk_bf16_f32 IS a fragment dword; hi back to fp32 by a shift / a mask
        u32x4 hi, lo;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                float t[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * h2 + e;
                    t[e] = fmaf(a.a[2], pn[rb][j][i], fmaf(a.a[1], pc[rb][j][i], a.a[0] * pm[rb][j][i]));
                }
                const unsigned hp = pack2(t[0], t[1]);
                hi[2 * rb + h2] = hp;
                if (SPLIT) {
                    const float r0 = t[0] - __builtin_bit_cast(float, hp << 16);
                    const float r1 = t[1] - __builtin_bit_cast(float, hp & 0xffff0000u);
                    lo[2 * rb + h2] = pack2(r0, r1);
                }
            }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            f32x4 acc = {0, 0, 0, 0};
            if (SPLIT) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo), hy[ib], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), hy[ib], acc, 0, 0, 0);
            u[ib] = acc * a.scale;
        }
    };

    auto step = [&](const int s, auto phase_tag) {
        constexpr int PH = decltype(phase_tag)::value;  // s mod 3 = ring slot of plane s = rotation slot
        constexpr int PC = (PH + 2) % 3, PM = (PH + 1) % 3;  // slots of planes s-1, s-2
        // "plane s has landed": at most the loads of the (kDepth - 1) younger planes may be outstanding.  Counting the
        // stores issued in between as well -- vmcnt((kDepth - 1) x 12), i.e. assuming that a younger store never
        // completes before an older LDS-DMA load -- produced a wrong plane about once in 400 runs (tools/dbg_mfma3.py);
        // loads complete in order among themselves, which is all this count relies on.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kDepth - 1) * kLoadsPerStep) : "memory");
        // ---- level 1: input plane s -> level-1 plane s - 1 (interior plane k0 - 3 + s) ----
        xpass(ring + PH * kSlot, p1[PH]);
        const int kz1 = k0 - 3 + s;
        const bool plane_in = kz1 >= 0 && kz1 < a.h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 u[2];
            zy(p1[PM], p1[PC], p1[PH], j, u);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const bool rin = plane_in && l1_row_in[ib];
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (rin && ((l1_col_in[j] >> i) & 1u)) ? u[ib][i] : 0.0f;
                u32x2 w;
                w.x = pack2(v[0], v[1]);
                w.y = pack2(v[2], v[3]);
                *reinterpret_cast<u32x2 *>(tile1 + (16 * ib + n16) * kPitch + 16 * j + 4 * g4) = w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- level 2: level-1 plane s - 1 -> output plane s - 2 ... (interior plane k0 - 4 + s) ----
        xpass(tile1, p2[PH]);
        const int kz2 = k0 - 4 + s;
        const bool live = kz2 >= k0 && kz2 < k_end;
        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
            a.out + (long) (max(kz2, 0) + 1) * a.plane, 0, live ? plane_bytes : 0u, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 u[2];
            zy(p2[PM], p2[PC], p2[PH], j, u);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                u32x2 w;
                w.x = pack2(u[ib][0], u[ib][1]);
                w.y = pack2(u[ib][2], u[ib][3]);
                __builtin_amdgcn_raw_buffer_store_b64(w, dst, store_off[j][ib], 0, 0);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        issue(s + kDepth, (PH + kDepth) % kRing);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nsteps = a.zc + 4;  // uniform across the launch (shorter last chunks run masked steps)
    for (int s = 0; s < nsteps; s += 3) {
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
        step(s + 2, std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

hipError_t launch_3d_bf16_mfma2(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    if (!p.mfma3_valid || p.boundary != LORA_BC_REFERENCE) return hipErrorNotSupported;
    ArgsMfma3 a;
    a.in = static_cast<const u16 *>(in);
    a.out = static_cast<u16 *>(out);
    a.h = p.dims[0];
    a.m = p.dims[1];
    a.n = p.dims[2];
    a.ld = a.n + 8;
    a.plane = (long) (a.m + 4) * (a.n + 8);
    if (a.plane * 2 >= (1L << 31)) return hipErrorInvalidValue;  // per-plane descriptors, 32-bit offsets
    a.z_begin = begin;
    a.z_end = end;
    a.tiles_x = (a.n + kOC - 1) / kOC;
    a.tiles_y = (a.m + kOR - 1) / kOR;
    int zc = p.fused_z_chunk;
    if (zc <= 0) {  // every chunk re-reads 4 planes: long chunks while a few rounds of 8 waves per CU remain
        zc = 64;
        const long per_plane = (long) a.tiles_x * a.tiles_y;
        while (zc > 8 && per_plane * ((end - begin + zc - 1) / zc) < 3 * 2048) zc /= 2;
    }
    zc = ((zc + 4 + 2) / 3) * 3 - 4;  // zc + 4 steps, a multiple of the 3-fold unrolled plane loop
    if (zc < 2) zc = 2;
    a.zc = zc;
    a.scale = p.mfma3_scale;
    for (int k = 0; k < 3; ++k) {
        a.c[k] = p.mfma3_abc[k];
        a.b[k] = p.mfma3_abc[3 + k];
        a.a[k] = p.mfma3_abc[6 + k];
    }
    const long chunks = ((long) end - begin + zc - 1) / zc;
    const long waves = chunks * a.tiles_x * a.tiles_y;
    const long nblocks = (waves + 3) / 4;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    if (p.mfma_split)
        hipLaunchKernelGGL(stencil3d_bf16_mfma2_kernel<true>, dim3((unsigned) nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(stencil3d_bf16_mfma2_kernel<false>, dim3((unsigned) nblocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

const char *kernel_name_3d_bf16_mfma2(const Plan &) { return "stencil3d_bf16_mfma2_kernel"; }

}  // namespace lora
