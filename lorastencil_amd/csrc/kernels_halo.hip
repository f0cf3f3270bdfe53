// kernels_halo.hip -- halo-cell kernels of the time-step driver (O(surface) work, never on the sweep's critical path).
//
// The reference never writes halo cells (2d/gpu.cu:266-271), which gives it the alternating boundary condition of
// SURVEY B2.  The driver here needs three halo operations on a padded array:
//   COPY  dst halo <- src halo   (temporal fusion keeps the level-0 halo in both buffers; Dirichlet boundary)
//   ZERO  dst halo <- 0          (restore the state of the reference's second buffer)
//   WRAP  dst halo <- the opposite interior edge of dst itself (periodic boundary, all dimensions at once:
//                                 every halo cell reads the interior cell it is a periodic image of)
// Halo widths follow the reference layouts: 4 (1D), 4 x 4 (2D), 1 x 2 x 4 (3D, 3d/main.cu:21-23).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "engine.h"

namespace lora {

typedef double d2v __attribute__((ext_vector_type(2)));

namespace {

struct HaloArgs {
    int nd;
    long ext[3];   // interior extents, outermost first (unused dims = 1)
    long halo[3];  // halo width per dimension (unused dims = 0)
    long region_end[6];
};

// Decode the k-th halo cell of a padded box into padded coordinates.  Regions, outermost dimension first:
// low and high slabs of dim 0 (full cross-section), then within the dim-0 interior the low/high slabs of dim 1
// (full dim-2 width), then within the dim-0/1 interior the low/high strips of dim 2.
__device__ __forceinline__ bool decode(const HaloArgs &a, long k, long c[3]) {
    const long P0 = a.ext[0] + 2 * a.halo[0], P1 = a.ext[1] + 2 * a.halo[1], P2 = a.ext[2] + 2 * a.halo[2];
    (void) P0;
    // region 0: dim-0 slabs: 2*halo0 x P1 x P2
    const long r0 = 2 * a.halo[0] * P1 * P2;
    if (k < r0) {
        const long s = k / (P1 * P2), rem = k - s * (P1 * P2);
        c[0] = s < a.halo[0] ? s : a.ext[0] + s;  // s in [halo0, 2 halo0) -> padded ext0 + s
        c[1] = rem / P2;
        c[2] = rem - c[1] * P2;
        return true;
    }
    k -= r0;
    // region 1: dim-1 slabs inside the dim-0 interior: ext0 x 2*halo1 x P2
    const long r1 = a.ext[0] * 2 * a.halo[1] * P2;
    if (k < r1) {
        const long i0 = k / (2 * a.halo[1] * P2), rem = k - i0 * (2 * a.halo[1] * P2);
        const long s = rem / P2;
        c[0] = i0 + a.halo[0];
        c[1] = s < a.halo[1] ? s : a.ext[1] + s;
        c[2] = rem - s * P2;
        return true;
    }
    k -= r1;
    // region 2: dim-2 strips inside the dim-0/1 interior: ext0 x ext1 x 2*halo2
    const long r2 = a.ext[0] * a.ext[1] * 2 * a.halo[2];
    if (k < r2) {
        const long w = 2 * a.halo[2];
        const long i0 = k / (a.ext[1] * w), rem = k - i0 * (a.ext[1] * w);
        const long i1 = rem / w, s = rem - i1 * w;
        c[0] = i0 + a.halo[0];
        c[1] = i1 + a.halo[1];
        c[2] = s < a.halo[2] ? s : a.ext[2] + s;
        return true;
    }
    return false;
}

template <typename T>
__global__ void halo_kernel(T *__restrict__ dst, const T *__restrict__ src, const HaloArgs a, int mode, long total) {
    const long P1 = a.ext[1] + 2 * a.halo[1], P2 = a.ext[2] + 2 * a.halo[2];
    for (long k = (long) blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long) gridDim.x * blockDim.x) {
        long c[3];
        if (!decode(a, k, c)) continue;
        const long off = (c[0] * P1 + c[1]) * P2 + c[2];
        if (mode == HALO_ZERO) {
            dst[off] = T(0);
        } else if (mode == HALO_COPY) {
            dst[off] = src[off];
        } else {  // HALO_WRAP: the interior cell this halo cell is the periodic image of
            long w[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                long x = c[d];
                if (x < a.halo[d])
                    x += a.ext[d];
                else if (x >= a.halo[d] + a.ext[d])
                    x -= a.ext[d];
                w[d] = x;
            }
            dst[off] = dst[(w[0] * P1 + w[1]) * P2 + w[2]];
        }
    }
}

}  // namespace

hipError_t launch_halo(const Plan &p, void *dst, const void *src, int mode, hipStream_t s) {
    HaloArgs a;
    a.nd = p.ndim;
    // map the shape's dims onto a 3-level box: unused leading dimensions have extent 1 and no halo
    const long h1[1] = {4}, h2[2] = {4, 4}, h3[3] = {1, 2, 4};
    const long *h = p.ndim == 1 ? h1 : (p.ndim == 2 ? h2 : h3);
    for (int d = 0; d < 3; ++d) {
        const int sd = d - (3 - p.ndim);  // index into the shape's own dims
        a.ext[d] = sd >= 0 ? p.dims[sd] : 1;
        a.halo[d] = sd >= 0 ? h[sd] : 0;
    }
    if (mode == HALO_WRAP)
        for (int d = 0; d < 3; ++d)
            if (a.halo[d] > a.ext[d]) return hipErrorInvalidValue;  // the wrap source would be a halo cell itself
    const long P1 = a.ext[1] + 2 * a.halo[1], P2 = a.ext[2] + 2 * a.halo[2];
    const long total = 2 * a.halo[0] * P1 * P2 + a.ext[0] * 2 * a.halo[1] * P2 + a.ext[0] * a.ext[1] * 2 * a.halo[2];
    if (total <= 0) return hipSuccess;
    const long want = (total + 255) / 256;
    const int blocks = (int) (want > 4096 ? 4096 : want);
    if (p.dtype == LORA_BF16)
        hipLaunchKernelGGL(halo_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, static_cast<unsigned short *>(dst),
                           static_cast<const unsigned short *>(src), a, mode, total);
    else
        hipLaunchKernelGGL(halo_kernel<double>, dim3(blocks), dim3(256), 0, s, static_cast<double *>(dst),
                           static_cast<const double *>(src), a, mode, total);
    return hipGetLastError();
}

// A rows x cols block of 8-byte elements between two strided arrays: the pack / unpack of a block decomposition's column
// ghost zones (lorastencil_amd/blocks.py).  A slab's ghost rows are one contiguous piece the sweep kernels read
// in place; a block's ghost COLUMNS are `cols` elements out of every row.  One lane per 16 bytes where both sides allow
// it (even cols, even leading dimensions and offsets: always the case for the driver's even ghost widths), else 8.
namespace {
template <typename V>
__global__ void copy_block_kernel(V *__restrict__ dst, long dst_ld, const V *__restrict__ src, long src_ld, long rows, long cols) {
    const long total = rows * cols;
    for (long k = (long) blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long) gridDim.x * blockDim.x) {
        const long r = k / cols, c = k - r * cols;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}
}  // namespace

hipError_t launch_copy_block(double *dst, long dst_ld, const double *src, long src_ld, long rows, long cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const bool wide = !((cols | dst_ld | src_ld) & 1) && !((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15);
    const long total = wide ? rows * cols / 2 : rows * cols;
    const long want = (total + 255) / 256;
    const int blocks = (int) (want > 8192 ? 8192 : want);
    if (wide)
        hipLaunchKernelGGL(copy_block_kernel<d2v>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<d2v *>(dst), dst_ld / 2,
                           reinterpret_cast<const d2v *>(src), src_ld / 2, rows, cols / 2);
    else
        hipLaunchKernelGGL(copy_block_kernel<double>, dim3(blocks), dim3(256), 0, s, dst, dst_ld, src, src_ld, rows, cols);
    return hipGetLastError();
}

// ---- the torus by ghost zones (capi.cpp: run_torus) -----------------------------------------------------------------
// Every cell of a padded array outside its interior becomes the periodic image of an interior cell: launch_halo's WRAP
// with any ring width (the pad plus the ghost zone of an extended grid), the interior given by its extents.
hipError_t launch_ring_wrap(int dtype, int nd, const int *dims, const int *ring, void *ptr, hipStream_t s) {
    HaloArgs a;
    a.nd = nd;
    for (int d = 0; d < 3; ++d) {
        const int sd = d - (3 - nd);
        a.ext[d] = sd >= 0 ? dims[sd] : 1;
        a.halo[d] = sd >= 0 ? ring[sd] : 0;
        if (a.halo[d] > a.ext[d]) return hipErrorInvalidValue;  // the wrap source would be a ring cell itself
    }
    const long P1 = a.ext[1] + 2 * a.halo[1], P2 = a.ext[2] + 2 * a.halo[2];
    const long total = 2 * a.halo[0] * P1 * P2 + a.ext[0] * 2 * a.halo[1] * P2 + a.ext[0] * a.ext[1] * 2 * a.halo[2];
    if (total <= 0) return hipSuccess;
    const long want = (total + 255) / 256;
    const int blocks = (int) (want > 8192 ? 8192 : want);
    if (dtype == LORA_BF16)
        hipLaunchKernelGGL(halo_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, static_cast<unsigned short *>(ptr),
                           static_cast<const unsigned short *>(nullptr), a, (int) HALO_WRAP, total);
    else
        hipLaunchKernelGGL(halo_kernel<double>, dim3(blocks), dim3(256), 0, s, static_cast<double *>(ptr),
                           static_cast<const double *>(nullptr), a, (int) HALO_WRAP, total);
    return hipGetLastError();
}

namespace {
// an e0 x e1 x e2 box of 8-byte units between two arrays with their own row and plane strides (in units)
__global__ void copy_box_kernel(double *__restrict__ dst, long dst_row, long dst_plane, const double *__restrict__ src, long src_row,
                                long src_plane, long e0, long e1, long e2) {
    const long total = e0 * e1 * e2;
    for (long k = (long) blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long) gridDim.x * blockDim.x) {
        const long i = k / (e1 * e2), rem = k - i * (e1 * e2), j = rem / e2, c = rem - j * e2;
        dst[i * dst_plane + j * dst_row + c] = src[i * src_plane + j * src_row + c];
    }
}
}  // namespace

// The interiors of two padded arrays of one shape and dtype, `dims` cells each, the arrays padded by `pad_dst` / `pad_src`
// per side (their halo, or halo + ghost zone): dst interior <- src interior.  Rows move as 8-byte units (bf16 rows are a
// multiple of 8 cells behind a pad that is a multiple of 4: whole units).
hipError_t launch_copy_interior(int dtype, int nd, const int *dims, void *dst, const int *pad_dst, const void *src, const int *pad_src,
                                hipStream_t s) {
    long e[3] = {1, 1, 1}, pd[3] = {0, 0, 0}, ps[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d) {
        const int sd = d - (3 - nd);
        if (sd >= 0) {
            e[d] = dims[sd];
            pd[d] = pad_dst[sd];
            ps[d] = pad_src[sd];
        }
    }
    const long per = dtype == LORA_BF16 ? 4 : 1;  // cells per 8-byte unit
    if (e[2] % per || pd[2] % per || ps[2] % per) return hipErrorInvalidValue;
    const long drow = (e[2] + 2 * pd[2]) / per, srow = (e[2] + 2 * ps[2]) / per;
    const long dplane = drow * (e[1] + 2 * pd[1]), splane = srow * (e[1] + 2 * ps[1]);
    double *d0 = static_cast<double *>(dst) + pd[0] * dplane + pd[1] * drow + pd[2] / per;
    const double *s0 = static_cast<const double *>(src) + ps[0] * splane + ps[1] * srow + ps[2] / per;
    const long total = e[0] * e[1] * (e[2] / per);
    if (total <= 0) return hipSuccess;
    const long want = (total + 255) / 256;
    const int blocks = (int) (want > 16384 ? 16384 : want);
    hipLaunchKernelGGL(copy_box_kernel, dim3(blocks), dim3(256), 0, s, d0, drow, dplane, s0, srow, splane, e[0], e[1], e[2] / per);
    return hipGetLastError();
}

}  // namespace lora
