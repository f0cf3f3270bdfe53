// Prototype (interior cells only, star taps): K applications of a 3D radius-1 stencil per launch with the time levels
// held in REGISTERS -- a lane owns 4 rows x 2 columns of a 4 NW x 128 tile at every level, x-neighbours come from the
// neighbouring lanes (v_mov_b32_dpp wave_shr / wave_shl), y-neighbours across waves through two LDS rows per wave and
// level, z by streaming planes through two partial sums per level and point.  Times the idea before it is built out
// (halo semantics, chunking, the separable box): prints GStencils/s for the interior of a 512^3 / 768^3 grid and checks a
// window against a CPU evaluation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bin/lanes3d_probe lanes3d_probe.hip && ./bin/lanes3d_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

struct Args {
    const double *in;
    double *out;
    int h, m, n;      // interior extents; padded array (h + 2) x (m + 4) x (n + 8)
    int ld;
    long plane;
    int zc, chunks, tiles_x, tiles_y;
    double wc, wx, wy, wz;  // star taps: centre, x-neighbours, y-neighbours, z-neighbours
};

__device__ __forceinline__ double dpp_shr(double v) {  // lane i <- lane i - 1
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) b, 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned) lo);
}
__device__ __forceinline__ double dpp_shl(double v) {  // lane i <- lane i + 1
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) b, 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned) lo);
}

template <int K, int NW>
__global__ __launch_bounds__(NW * 64, (NW > 8 ? 3 : 2)) void lanes3d(const Args a) {
    constexpr int OW = 120, OH = 4 * NW - 2 * K;
    __shared__ __attribute__((aligned(16))) double rows[K][NW][2][128];  // level l - 1: wave's top / bottom row
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = a.tiles_x * a.tiles_y;
    const int chunk = blockIdx.x / per, rem = blockIdx.x - chunk * per;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int k0 = chunk * a.zc;
    const int zc = min(a.zc, a.h - k0);
    const int X0 = tx * OW - 4, Y0 = ty * OH - K;  // interior coordinates of the tile's first column / row
    // this lane's block: rows Y0 + 4 wv + r, columns X0 + 2 lane, +1   (padded: +2 rows, +4 columns), clamped
    const int pc = min(max(X0 + 2 * lane + 4, 0), a.n + 6);
    long off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) off[r] = (long) min(max(Y0 + 4 * wv + r + 2, 0), a.m + 3) * a.ld + pc;
    const int up = max(wv - 1, 0), dn = min(wv + 1, NW - 1);

    double a1[K][4][2], a0[K][4][2];
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[l][r][0] = a1[l][r][1] = a0[l][r][0] = a0[l][r][1] = 0.0;
    d2 nxt[4], outv[4];
    auto load_plane = [&](int p) {
        const double *src = a.in + (long) min(max(k0 - K + p + 1, 0), a.h + 1) * a.plane;
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nxt[r]) : "v"(src + off[r]) : "memory");
    };
    const bool st_lane = lane >= 2 && lane < 62 && X0 + 2 * lane < a.n;
    load_plane(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) outv[r] = (d2){0.0, 0.0};
    const int steps = zc + 2 * K + 1;
    for (int p = 0; p < steps; ++p) {
        // everything issued one step ago has had a whole step to complete: the loads of plane p, the stores of plane p - 2
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]) : : "memory");
        {
            const int o = p - 2 * K - 1;  // output plane completed in the previous step
            if (o >= 0 && o < zc && st_lane) {
                double *dst = a.out + (long) (k0 + o + 1) * a.plane;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * wv + r;
                    if (row >= K && row < 4 * NW - K && Y0 + row < a.m) *reinterpret_cast<d2 *>(dst + off[r]) = outv[r];
                }
            }
        }
        double v[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r][0] = nxt[r].x;
            v[r][1] = nxt[r].y;
        }
        load_plane(p + 1);
#pragma unroll
        for (int l = 0; l < K; ++l) {
            // level l + 1 from the plane of level l in v: boundary rows through LDS
            *reinterpret_cast<d2 *>(&rows[l][wv][0][2 * lane]) = (d2){v[0][0], v[0][1]};
            *reinterpret_cast<d2 *>(&rows[l][wv][1][2 * lane]) = (d2){v[3][0], v[3][1]};
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const d2 vu = *reinterpret_cast<const d2 *>(&rows[l][up][1][2 * lane]);
            const d2 vd = *reinterpret_cast<const d2 *>(&rows[l][dn][0][2 * lane]);
            double nv[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double xl = dpp_shr(v[r][1]), xr = dpp_shl(v[r][0]);
                const double u0 = r == 0 ? vu.x : v[r - 1][0], u1 = r == 0 ? vu.y : v[r - 1][1];
                const double b0 = r == 3 ? vd.x : v[r + 1][0], b1 = r == 3 ? vd.y : v[r + 1][1];
                // dz = 2 tap completes the plane below; dz = 1 taps in (dy, dx) order; dz = 0 tap opens the plane above
                nv[r][0] = fma(a.wz, v[r][0], a1[l][r][0]);
                nv[r][1] = fma(a.wz, v[r][1], a1[l][r][1]);
                double s0 = fma(a.wy, u0, a0[l][r][0]), s1 = fma(a.wy, u1, a0[l][r][1]);
                s0 = fma(a.wx, xl, s0);
                s1 = fma(a.wx, v[r][0], s1);
                s0 = fma(a.wc, v[r][0], s0);
                s1 = fma(a.wc, v[r][1], s1);
                s0 = fma(a.wx, v[r][1], s0);
                s1 = fma(a.wx, xr, s1);
                a1[l][r][0] = fma(a.wy, b0, s0);
                a1[l][r][1] = fma(a.wy, b1, s1);
                a0[l][r][0] = fma(a.wz, v[r][0], 0.0);
                a0[l][r][1] = fma(a.wz, v[r][1], 0.0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r][0] = nv[r][0];
                v[r][1] = nv[r][1];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) outv[r] = (d2){v[r][0], v[r][1]};
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]) : : "memory");
}

template <int K, int NW>
double run(int h, int m, int n, int zc, bool check) {
    const size_t count = (size_t) (h + 2) * (m + 4) * (n + 8);
    std::vector<double> hin(count);
    for (size_t i = 0; i < count; ++i) hin[i] = (double) ((i * 2654435761u) % 97);
    double *din, *dout;
    hipMalloc(&din, count * 8);
    hipMalloc(&dout, count * 8);
    hipMemcpy(din, hin.data(), count * 8, hipMemcpyHostToDevice);
    hipMemset(dout, 0, count * 8);
    Args a;
    a.in = din;
    a.out = dout;
    a.h = h;
    a.m = m;
    a.n = n;
    a.ld = n + 8;
    a.plane = (long) (m + 4) * (n + 8);
    a.zc = zc;
    a.chunks = (h + zc - 1) / zc;
    a.tiles_x = (n + 119) / 120;
    a.tiles_y = (m + (4 * NW - 2 * K) - 1) / (4 * NW - 2 * K);
    a.wc = 2.0;
    a.wx = a.wy = a.wz = 1.0;
    const int nb = a.chunks * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL((lanes3d<K, NW>), dim3(nb), dim3(NW * 64), 0, 0, a);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lanes3d<K, NW>), dim3(nb), dim3(NW * 64), 0, 0, a);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double gst = (double) h * m * n * K / (us * 1e-6) / 1e9;
    printf("K=%d NW=%d %dx%dx%d zc=%d blocks=%d: %.1f us per launch, %.1f GStencils/s (interior cells only are right)\n", K, NW, h, m, n, zc, nb, us, gst);
    if (check) {
        // CPU: K sweeps of the whole grid (halo = the input's own), compared K + 1 cells away from every face
        std::vector<double> hout(count), cur(hin), nx(hin);
        hipMemcpy(hout.data(), dout, count * 8, hipMemcpyDeviceToHost);
        auto idx = [&](int z, int y, int x) { return (size_t) (z + 1) * a.plane + (size_t) (y + 2) * a.ld + x + 4; };
        for (int k = 1; k <= K; ++k) {
            for (int z = 0; z < h; ++z)
                for (int y = 0; y < m; ++y)
                    for (int x = 0; x < n; ++x)
                        nx[idx(z, y, x)] = 2.0 * cur[idx(z, y, x)] + cur[idx(z - 1, y, x)] + cur[idx(z + 1, y, x)] + cur[idx(z, y - 1, x)] +
                                           cur[idx(z, y + 1, x)] + cur[idx(z, y, x - 1)] + cur[idx(z, y, x + 1)];
            std::swap(cur, nx);
        }
        long bad = 0, total = 0;
        int first[3] = {-1, -1, -1};
        for (int z = K + 1; z < h - K - 1; ++z)
            for (int y = K + 1; y < m - K - 1; ++y)
                for (int x = K + 1; x < n - K - 1; ++x) {
                    ++total;
                    if (hout[idx(z, y, x)] != cur[idx(z, y, x)]) {
                        if (!bad) first[0] = z, first[1] = y, first[2] = x;
                        ++bad;
                    }
                }
        printf("   deep-interior check: %s (%ld of %ld differ; first at z=%d y=%d x=%d)\n", bad ? "MISMATCH" : "ok", bad, total, first[0], first[1], first[2]);
    }
    hipFree(din);
    hipFree(dout);
    return gst;
}

int main() {
    run<4, 8>(70, 100, 260, 30, true);
    run<3, 8>(70, 100, 260, 30, true);
    run<2, 8>(70, 100, 260, 30, true);
    run<4, 8>(512, 512, 512, 256, false);
    run<3, 8>(512, 512, 512, 256, false);
    run<4, 8>(768, 768, 768, 768, false);
    return 0;
}
