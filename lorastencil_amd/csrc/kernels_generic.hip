// kernels_generic.hip -- fallback sweeps for grids the tiled kernels cannot take: an ODD innermost extent makes the
// padded row length odd, so rows are only 8-byte aligned and the 16-byte row pieces of the tiled kernels do not
// exist.  One thread per interior point, scalar 8-byte accesses (neighbouring lanes share cache lines through L1/L2),
// taps in the reference's order with fused multiply-adds.  Correct for every size and every tap table; not tuned --
// the reference itself reads and writes out of bounds for sizes off its tile grid.
#include <hip/hip_runtime.h>

#include "engine.h"

namespace lora {

namespace {

__global__ __launch_bounds__(256) void stencil2d_generic_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                                 int m, int n, int row_begin, int row_end,
                                                                 const Taps49 W) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = row_begin + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= n || i >= row_end) return;
    const long ld = n + 8;
    const double *c = in + (long) (i + 4) * ld + (j + 4);
    double s = 0.0;
#pragma unroll
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) {
            const double w = W.w[dy * 7 + dx];
            if (w != 0.0) s = fma(w, c[(dy - 3) * ld + (dx - 3)], s);
        }
    out[(long) (i + 4) * ld + (j + 4)] = s;
}

__global__ __launch_bounds__(256) void stencil3d_generic_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                                 int h, int m, int n, int z_begin, int z_end,
                                                                 const Taps27 W) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int k = z_begin + blockIdx.z;
    if (j >= n || i >= m || k >= z_end) return;
    const long ld = n + 8, plane = (long) (m + 4) * ld;
    const double *c = in + (long) (k + 1) * plane + (long) (i + 2) * ld + (j + 4);
    double s = 0.0;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const double w = W.w[dz * 9 + dy * 3 + dx];
                if (w != 0.0) s = fma(w, c[(dz - 1) * plane + (dy - 1) * ld + (dx - 1)], s);
            }
    out[(long) (k + 1) * plane + (long) (i + 2) * ld + (j + 4)] = s;
}

}  // namespace

hipError_t launch_2d_generic(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    Taps49 w;
    for (int k = 0; k < 49; ++k) w.w[k] = p.w[k];
    const dim3 grid((p.dims[1] + 63) / 64, (end - begin + 3) / 4);
    if (grid.y > 65535u) {
        // split tall launches: grid.y is limited to 65535
        const int rows = 65535 * 4;
        for (int b = begin; b < end; b += rows) {
            const int e = b + rows < end ? b + rows : end;
            hipLaunchKernelGGL(stencil2d_generic_kernel, dim3(grid.x, (e - b + 3) / 4), dim3(256), 0, s, in, out, p.dims[0],
                               p.dims[1], b, e, w);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(stencil2d_generic_kernel, grid, dim3(256), 0, s, in, out, p.dims[0], p.dims[1], begin, end, w);
    return hipGetLastError();
}

hipError_t launch_3d_generic(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s) {
    if (end <= begin) return hipSuccess;
    Taps27 w;
    for (int k = 0; k < 27; ++k) w.w[k] = p.w[k];
    for (int b = begin; b < end; b += 65535) {
        const int e = b + 65535 < end ? b + 65535 : end;
        const dim3 grid((p.dims[2] + 63) / 64, (p.dims[1] + 3) / 4, e - b);
        if (grid.y > 65535u) return hipErrorInvalidValue;
        hipLaunchKernelGGL(stencil3d_generic_kernel, grid, dim3(256), 0, s, in, out, p.dims[0], p.dims[1], p.dims[2], b, e,
                           w);
    }
    return hipGetLastError();
}

const char *kernel_name_generic(const Plan &p) {
    return p.ndim == 2 ? "stencil2d_generic_kernel" : "stencil3d_generic_kernel";
}

}  // namespace lora
