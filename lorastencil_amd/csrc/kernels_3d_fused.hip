// kernels_3d_fused.hip -- TWO applications of a 3D radius-1 stencil per launch (temporal fusion, fp64).
//
// The single-sweep kernel (kernels_3d.hip) streams the compulsory 16 B per point and application and sits at what
// HBM delivers (~5 TB/s).  Here one launch reads the grid once, applies the stencil twice and writes once: per
// workgroup the first application's result (time level 1) of plane q lives only in LDS, for as long as it takes to
// add it to the three level-2 planes it touches.  The reference has no temporal blocking.
//
// Semantics = two consecutive launches of the reference driver starting at an EVEN step: level 1 is
// "buffer 1", whose halo cells are never written and hold 0, so every level-1 cell outside the interior is 0 (under
// the Dirichlet option: the source's halo value); the
// input's halo is whatever the source buffer holds (the driver in capi.cpp keeps the level-0 halo in both physical
// buffers while fused launches run, exactly as for the 2D fused kernel).  Taps are applied in the single-sweep
// kernel's order at both levels, so the result is bit-identical to two single sweeps.
//
// Geometry (256 threads; a wave = 2 row groups x 32 lanes, a lane owns 2 adjacent columns x RY rows -> 8 strips):
//   level-2 (output) tile   OH = 8 RY - 2 rows x  60 columns   (30 lanes x 2; tile origin J = 60 tx is even)
//   level-1 tile            8 RY rows        x  64 columns   in LDS (B), starts 1 row / 2 columns before the output
//   input window            8 RY + 2 rows    x  68 columns   in LDS (A), starts 2 rows / 4 columns before: an even
//                                                            padded column, so every global load is an aligned 16 B
// Both levels use the same lane -> window map (3 x ds_read_b128 per row, taps at window elements 1..4), the same
// three rotating register accumulator sets per level (input plane p feeds level-1 planes p+1, p, p-1; level-1 plane
// q feeds output planes q+1, q, q-1) and the same 3-way unrolled plane loop as the single-sweep kernel.  One plane
// iteration = level-1 scatter from A, publish the completed level-1 plane in B, barrier, level-2 scatter from B,
// store the completed output plane, refill A from the registers that prefetched the next input plane, barrier.
// A and B are single-buffered (37 KB per workgroup).  A chunk of zc output planes consumes zc + 4 input planes.
// Cost of fusing: the level-1 tile is (8 RY x 64) / (OH x 60) = 1.14 x the output tile, ragged grid edges aside.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "planes_3d.h"

namespace lora {

namespace {

constexpr int kLanesX = 32;                  // lanes per row group
constexpr int kGroups = 64 / kLanesX;        // row groups per wave
constexpr int kStrips = 4 * kGroups;         // strips per workgroup
constexpr int kMidW = 2 * kLanesX;           // level-1 columns per tile
constexpr int kOutW = kMidW - 4;             // output columns per tile
constexpr int kInW = kMidW + 4;              // staged input columns (also the LDS row stride of both tiles)
constexpr int kInChunks = kInW / 2;          // 16-byte pieces per staged row

struct ArgsF3 {
    const double *in;
    double *out;
    int h, m, n;
    int ld;
    long plane;
    int z_begin, z_end;
    int zc;
    int tiles_x, tiles_y;
    int dirichlet;  // level-1 cells outside the interior keep the source's halo value instead of 0
};

// DIRICHLET is a template flag, not a run-time one: the halo reload it adds to the publish step costs registers that
// the reference-boundary instantiation should not pay (bf16: 137 instead of 128 VGPRs = 3 instead of 4 per CU).
template <int TAPSET, int RY, bool DIRICHLET>
__global__ __launch_bounds__(256, 3) void stencil3d_fused2_kernel(const ArgsF3 a, const Taps27 W) {
    constexpr int MH = kStrips * RY;  // level-1 rows
    constexpr int OH = MH - 2;        // output rows
    constexpr int IH = MH + 2;        // input rows (B is given the same height: the last strip reads 2 rows past MH)
    constexpr int NCHUNK = IH * kInChunks;
    constexpr int NIT = (NCHUNK + 255) / 256;
    __shared__ __attribute__((aligned(16))) double A[IH * kInW];
    __shared__ __attribute__((aligned(16))) double B[IH * kInW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int sid = wv * kGroups + lane / kLanesX;  // strip: level-1 rows sid*RY .., output rows sid*RY ..
    const int cl = lane % kLanesX;                  // column pair inside the tile

    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int per_chunk = a.tiles_x * a.tiles_y;
    const int chunk = lin / per_chunk;
    const int rem = lin - chunk * per_chunk;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;  // first output plane (interior index) of the chunk
    const int I = ty * OH;                    // first output row
    const int J = tx * kOutW;                 // first output column (even)
    const int zc = min(a.zc, a.z_end - k0);
    const int nplanes = zc + 4;               // input planes: interior k0-2 .. k0+zc+1

    // staging coordinates: tile row r / piece c <-> padded row I + r, padded column J + 2c (clamped inside the array;
    // clamped pieces only ever feed level-1 cells that are outside the interior and forced to 0)
    int goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * 256;
        const int r = k / kInChunks;
        const int c = k - r * kInChunks;
        const int gr = min(I + r, a.m + 3);
        const int gc = min(J + 2 * c, a.n + 6);
        goff[it] = gr * a.ld + gc;
    }
    d2 stage[NIT];
    auto load_plane = [&](int p) {
        const double *src = a.in + (long) min(max(k0 - 1 + p, 0), a.h + 1) * a.plane;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (NCHUNK % 256 == 0 || tid + it * 256 < NCHUNK) stage[it] = *reinterpret_cast<const d2 *>(src + goff[it]);
        }
    };
    auto write_plane = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = tid + it * 256;
            if (NCHUNK % 256 == 0 || k < NCHUNK) *reinterpret_cast<d2 *>(&A[2 * k]) = stage[it];
        }
    };

    double a0[3][RY], a1[3][RY];  // level-1 partial sums
    double b0[3][RY], b1[3][RY];  // level-2 partial sums
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < RY; ++r) a0[s][r] = a1[s][r] = b0[s][r] = b1[s][r] = 0.0;

    // level-1 cell of this lane: interior row I - 1 + sid*RY + r, interior columns J - 2 + 2 cl (+1)
    const int col1 = J - 2 + 2 * cl;
    const bool col1_in = col1 >= 0 && col1 < a.n;  // n is even: both columns of the pair fall on the same side
    const int row1 = I - 1 + sid * RY;
    // output cell: interior row I + sid*RY + r, interior columns J + 2 cl (+1), lanes 0..29 only
    const int colo = J + 2 * cl;
    const bool colo_ok = cl < kLanesX - 2 && colo < a.n;
    const int rowo = I + sid * RY;
    const int strip_off = (sid * RY) * kInW + 2 * cl;
    double *const out_col = a.out + (long) (rowo + 2) * a.ld + (colo + 4);

    load_plane(0);
    write_plane();
    __syncthreads();

    auto consume = [&](int p, auto phase_tag) {
        constexpr int PH = decltype(phase_tag)::value;  // p mod 3
        constexpr int PH2 = (PH + 2) % 3;               // (p - 1) mod 3: phase of the level-1 plane finished below
        const bool more = p + 1 < nplanes;
        if (more) load_plane(p + 1);

        // ---- level 1: input plane p (interior k0-2+p) -> level-1 planes p+1, p, p-1 ----
        {
            const double *strip = &A[strip_off];
#pragma unroll
            for (int j = 0; j < RY + 2; ++j) {
                double win[6];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const d2 v = *reinterpret_cast<const d2 *>(strip + j * kInW + 2 * q);
                    win[2 * q] = v.x;
                    win[2 * q + 1] = v.y;
                }
                scatter_row<TAPSET, RY, PH>(a0, a1, win, j, W);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < RY; ++r) asm volatile("" : "+v"(a0[s][r]), "+v"(a1[s][r]));
        }
        // level-1 plane p-1 (interior k0-3+p) is complete in slot (PH - 2) mod 3: publish it, 0 outside the interior
        {
            constexpr int s = (PH + 1) % 3;
            const int z1 = k0 - 3 + p;
            const bool z_in = z1 >= 0 && z1 < a.h;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const bool in = z_in && col1_in && row1 + r >= 0 && row1 + r < a.m;
                d2 v;
                v.x = in ? a0[s][r] : 0.0;
                v.y = in ? a1[s][r] : 0.0;
                if (DIRICHLET && !in) {
                    // fixed boundary: a halo cell keeps the caller's value at every level -- read it from the source
                    // (only lanes on the grid's rim get here; cells beyond the padded array feed no valid output)
                    const int pz = z1 + 1, pr = row1 + r + 2, pc = col1 + 4;
                    if (pz >= 0 && pz <= a.h + 1 && pr >= 0 && pr <= a.m + 3 && pc >= 0 && pc + 1 <= a.n + 7)
                        v = *reinterpret_cast<const d2 *>(a.in + (long) pz * a.plane + (long) pr * a.ld + pc);
                }
                *reinterpret_cast<d2 *>(&B[strip_off + r * kInW]) = v;
                a0[s][r] = 0.0;
                a1[s][r] = 0.0;
            }
        }
        __syncthreads();

        // ---- level 2: level-1 plane p-1 -> output planes p, p-1, p-2 (same index convention) ----
        {
            const double *strip = &B[strip_off];
#pragma unroll
            for (int j = 0; j < RY + 2; ++j) {
                double win[6];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const d2 v = *reinterpret_cast<const d2 *>(strip + j * kInW + 2 * q);
                    win[2 * q] = v.x;
                    win[2 * q + 1] = v.y;
                }
                scatter_row<TAPSET, RY, PH2>(b0, b1, win, j, W);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < RY; ++r) asm volatile("" : "+v"(b0[s][r]), "+v"(b1[s][r]));
        }
        // output plane k0 + p - 4 is complete in slot (PH2 - 2) mod 3 = PH
        {
            constexpr int s = PH;
            const int o = p - 4;
            if (o >= 0 && o < zc && colo_ok) {
                double *dst = out_col + (long) (k0 + o + 1) * a.plane;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (sid * RY + r < OH && rowo + r < a.m) {
                        d2 v;
                        v.x = b0[s][r];
                        v.y = b1[s][r];
                        *reinterpret_cast<d2 *>(dst + (long) r * a.ld) = v;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                b0[s][r] = 0.0;
                b1[s][r] = 0.0;
            }
        }
        if (more) write_plane();
        __syncthreads();
    };

    for (int p = 0; p < nplanes; p += 3) {
        consume(p, std::integral_constant<int, 0>{});
        if (p + 1 < nplanes) consume(p + 1, std::integral_constant<int, 1>{});
        if (p + 2 < nplanes) consume(p + 2, std::integral_constant<int, 2>{});
    }
}

template <int TAPSET, int RY>
hipError_t launch_fused3(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    constexpr int OH = kStrips * RY - 2;
    ArgsF3 a;
    a.in = in;
    a.out = out;
    a.h = p.dims[0];
    a.m = p.dims[1];
    a.n = p.dims[2];
    a.ld = a.n + 8;
    a.plane = (long) (a.m + 4) * (a.n + 8);
    if (a.plane >= (1L << 31)) return hipErrorInvalidValue;  // 32-bit in-plane offsets
    a.z_begin = begin;
    a.z_end = end;
    a.dirichlet = p.boundary == LORA_BC_DIRICHLET;
    a.tiles_x = (a.n + kOutW - 1) / kOutW;
    a.tiles_y = (a.m + OH - 1) / OH;
    // every chunk re-reads 4 planes: long chunks while they still leave a few workgroups per CU slot
    int zc = p.fused_z_chunk;
    if (zc <= 0) {
        zc = 32;
        const long per_plane = (long) a.tiles_x * a.tiles_y;
        while (zc > 8 && per_plane * ((end - begin + zc - 1) / zc) < 6 * 768) zc /= 2;  // 768 = 3 per CU
    }
    a.zc = zc;
    const long chunks = ((long) end - begin + a.zc - 1) / a.zc;
    const long nblocks = chunks * a.tiles_x * a.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    Taps27 w;
    for (int k = 0; k < 27; ++k) w.w[k] = p.w[k];
    if (a.dirichlet)
        hipLaunchKernelGGL((stencil3d_fused2_kernel<TAPSET, RY, true>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    else
        hipLaunchKernelGGL((stencil3d_fused2_kernel<TAPSET, RY, false>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_3d_fused2(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    if (p.tapset == TAPS3D_STAR) return launch_fused3<TAPS3D_STAR, 4>(p, in, out, begin, end, s);
    return launch_fused3<TAPS3D_BOX, 4>(p, in, out, begin, end, s);
}

const char *kernel_name_3d_fused2(const Plan &) { return "stencil3d_fused2_kernel"; }

}  // namespace lora
