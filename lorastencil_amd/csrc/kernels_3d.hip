// kernels_3d.hip -- 3D radius-1 sweeps for gfx950 on a padded (h+2) x (m+4) x (n+8) fp64 grid (halos z 1, y 2,
// x 4: 3d/main.cu:21-23).  Replaces kernel_star3d1r (3d/gpu_star.cu:101-133) and kernel_box3d1r
// (3d/gpu_box.cu:105-140).
//
// The reference streams ALL planes through one (8 x 64)-tile workgroup: 512 workgroups at 512^3, with four
// block-wide barriers per plane and a shared-memory ring of per-plane partial results.  That cannot fill 256 CUs.
// Here the grid is also cut along z:
//   * a 256-thread workgroup owns a TY x 128 column (TY = 4 waves x RY rows) and a chunk of `zc` output planes; it
//     streams the zc+2 input planes of the chunk through a double-buffered LDS plane tile ((TY+2) x 136 doubles):
//     the 16-byte coalesced global loads of plane p+1 are in flight while plane p is consumed, one barrier per plane;
//   * each lane owns 2 adjacent columns x RY rows and keeps three rotating sets of output-plane accumulators in
//     registers: input plane q adds its in-plane 3x3 (box) or cross (star) sums, weighted per dz, to output planes
//     q+1, q, q-1 -- the register analogue of the reference's 3-slot ring, without the LDS round trips;
//   * per input row a lane reads a 6-wide window (3 x ds_read_b128, conflict-free) and reuses it for up to
//     3 rows x 3 planes of outputs;
//   * taps are applied dz-major, then dy, then dx: the order of the reference's CPU check (3d/main.cu:33-68), with
//     fused multiply-adds;
//   * output plane o is stored as soon as input plane o+2 has been consumed (16 bytes per lane, 1 KiB per wave);
//     halo cells are never written (3d/gpu_star.cu:122-127).
//   * blocks that share an XCD get a contiguous run of tiles so that x/y halos are re-read from that XCD's L2.
#include <hip/hip_runtime.h>

#include "device_common.h"

namespace lora {

namespace {


constexpr int kTileW = 128;
constexpr int kLdsW = kTileW + 8;
constexpr int kChunksPerRow = kLdsW / 2;

template <int TAPSET>
__host__ __device__ constexpr bool tap_on3(int dz, int dy, int dx) {
    return TAPSET == TAPS3D_BOX ? true : (((dz != 1) + (dy != 1) + (dx != 1)) <= 1);
}

struct Args3D {
    const double *in;
    double *out;
    int h, m, n;             // interior extents
    int ld;                  // padded row length n + 8
    long plane;              // padded plane size (m + 4) * (n + 8)
    int z_begin, z_end;      // interior plane range of this launch
    int zc;                  // output planes per workgroup
    int tiles_x, tiles_y;
};

template <int TAPSET, int RY>
__global__ __launch_bounds__(256, 4) void stencil3d_stream_kernel(const Args3D a, const Taps27 W) {
    constexpr int TY = 4 * RY;
    constexpr int LH = TY + 2;
    constexpr int NCHUNK = LH * kChunksPerRow;
    constexpr int NIT = (NCHUNK + 255) / 256;
    __shared__ __attribute__((aligned(16))) double tile[2][LH * kLdsW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;

    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int per_chunk = a.tiles_x * a.tiles_y;
    const int chunk = lin / per_chunk;
    const int rem = lin - chunk * per_chunk;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int k0 = a.z_begin + chunk * a.zc;  // first interior plane of the chunk
    const int i0 = ty * TY;
    const int j0 = tx * kTileW;
    const int zc = min(a.zc, a.z_end - k0);   // output planes this workgroup really owns
    const int nplanes = zc + 2;               // input planes: padded k0 .. k0+zc+1

    // per-thread staging coordinates (independent of the plane)
    long goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * 256;
        const int r = k / kChunksPerRow;
        const int c = k - r * kChunksPerRow;
        const int gr = min(i0 + 1 + r, a.m + 3);  // padded rows i0+1 .. i0+TY+2
        const int gc = min(j0 + 2 * c, a.n + 6);
        goff[it] = (long) gr * a.ld + gc;
    }
    d2 stage[NIT];
    auto load_plane = [&](int p) {
        const double *src = a.in + (long) min(k0 + p, a.h + 1) * a.plane;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (NCHUNK % 256 == 0 || tid + it * 256 < NCHUNK) stage[it] = *reinterpret_cast<const d2 *>(src + goff[it]);
        }
    };
    auto write_plane = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = tid + it * 256;
            if (NCHUNK % 256 == 0 || k < NCHUNK) *reinterpret_cast<d2 *>(&tile[buf][2 * k]) = stage[it];
        }
    };

    double acc0[3][RY], acc1[3][RY];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            acc0[s][r] = 0.0;
            acc1[s][r] = 0.0;
        }

    const int col = j0 + 2 * lane;
    const bool col_ok = col < a.n;
    const int strip_off = (wv * RY) * kLdsW + 2 * lane + 2;  // window = tile cols 2*lane+2 .. 2*lane+7
    double *const out_col = a.out + (long) (i0 + wv * RY + 2) * a.ld + (col + 4);

    load_plane(0);
    write_plane(0);
    __syncthreads();

    // PHASE = p mod 3 is a compile-time constant inside the 3-way unrolled body, so the accumulator set of
    // output plane o = p - dz is the static index (PHASE - dz) mod 3.
    auto consume = [&](int p, auto phase_tag) {
        constexpr int PHASE = decltype(phase_tag)::value;
        const bool more = p + 1 < nplanes;
        if (more) load_plane(p + 1);
        const double *strip = &tile[p & 1][strip_off];
#pragma unroll
        for (int j = 0; j < RY + 2; ++j) {
            double win[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const d2 v = *reinterpret_cast<const d2 *>(strip + j * kLdsW + 2 * q);
                win[2 * q] = v.x;
                win[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                constexpr int dummy = 0;
                (void) dummy;
                const int s = (PHASE - dz + 3) % 3;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const int dy = j - r;
                    if (dy >= 0 && dy < 3) {
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            if (tap_on3<TAPSET>(dz, dy, dx)) {
                                const double wt = W.w[dz * 9 + dy * 3 + dx];
                                acc0[s][r] = fma(wt, win[dx + 1], acc0[s][r]);
                                acc1[s][r] = fma(wt, win[dx + 2], acc1[s][r]);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // keep the partial sums where they are (no sinking into the predicated stores)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < RY; ++r) asm volatile("" : "+v"(acc0[s][r]), "+v"(acc1[s][r]));

        // output plane o = p - 2 is complete (it received dz = 0, 1, 2 from planes o, o+1, o+2)
        {
            constexpr int s = (PHASE - 2 + 3) % 3;
            const int o = p - 2;
            if (o >= 0 && o < zc && col_ok) {
                double *dst = out_col + (long) (k0 + o + 1) * a.plane;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (i0 + wv * RY + r < a.m) {
                        d2 v;
                        v.x = acc0[s][r];
                        v.y = acc1[s][r];
                        *reinterpret_cast<d2 *>(dst + (long) r * a.ld) = v;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                acc0[s][r] = 0.0;
                acc1[s][r] = 0.0;
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

template <int TAPSET, int RY>
hipError_t launch_stream(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    constexpr int TY = 4 * RY;
    Args3D a;
    a.in = in;
    a.out = out;
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
    Taps27 w;
    for (int k = 0; k < 27; ++k) w.w[k] = p.w[k];
    hipLaunchKernelGGL((stencil3d_stream_kernel<TAPSET, RY>), dim3((unsigned) nblocks), dim3(256), 0, s, a, w);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_3d(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (p.tapset == TAPS3D_STAR) return launch_stream<TAPS3D_STAR, 4>(p, in, out, begin, end, s);
    return launch_stream<TAPS3D_BOX, 4>(p, in, out, begin, end, s);
}

const char *kernel_name_3d(const Plan &) { return "stencil3d_stream_kernel"; }

}  // namespace lora
