// weights.cpp -- host side of the hot path that is not a kernel: the params tables of the
// harness, the params -> applied-taps mapping of each operator (including the low-rank factor
// precompute of the box2d operator) and the glibc-compatible input fill.
//
// Reference behaviour followed (file:line under /root/reference/src/):
//   params tables          1d/main.cu:77-78, 2d/main.cu:139-195, 3d/main.cu:112-125
//   pyramid factorisation  2d/gpu.cu:280-350
//   which params are used  1d/gpu_1r.cu:92-99, 2d/gpu.cu:430-444, :486-487, 3d/gpu_star.cu:142-151,
//                          3d/gpu_box.cu:151-164
//   rand() fill            1d/main.cu:105-109, 2d/main.cu:232-236, 3d/main.cu:164-168
#include <algorithm>
#include <cmath>
#include <cstring>

#include "engine.h"

namespace lora {

int shape_ndim(int shape) {
    switch (shape) {
        case LORA_1D1R:
        case LORA_1D2R:
            return 1;
        case LORA_STAR2D1R:
        case LORA_BOX2D1R:
        case LORA_STAR2D3R:
        case LORA_BOX2D3R:
            return 2;
        case LORA_STAR3D1R:
        case LORA_BOX3D1R:
            return 3;
        default:
            return 0;
    }
}

int shape_ntaps(int shape) {
    static const int taps[4] = {0, 9, 49, 27};
    return taps[shape_ndim(shape)];
}

// ---- 7x7 helpers -------------------------------------------------------------------------
namespace {

struct Mat7 {
    double a[7][7];
    Mat7() { std::memset(a, 0, sizeof(a)); }
    explicit Mat7(const double *p) { std::memcpy(a, p, sizeof(a)); }
};

// Set the eight symmetric images of (i, j) (offsets from the centre) to v.
void set_symmetric8(Mat7 &m, int i, int j, double v) {
    const int ii[2] = {3 + i, 3 - i}, jj[2] = {3 + j, 3 - j};
    for (int a : ii)
        for (int b : jj) {
            m.a[a][b] = v;
            m.a[b][a] = v;
        }
}

}  // namespace

int default_params(int shape, double *p) {
    switch (shape) {
        case LORA_1D1R:
        case LORA_1D2R: {
            // triangle 0..4..0 (1d1r) or 1..5..1 (1d2r)
            const int base = (shape == LORA_1D1R) ? 0 : 1;
            for (int t = 0; t < 9; ++t) p[t] = base + 4 - std::abs(t - 4);
            return 9;
        }
        case LORA_STAR2D1R: {
            // (0,1,2,4,2,1,0) (x) itself, +1 on the four axis tips, -1 on the four (+-2,+-2) corners
            // == the literal table at 2d/main.cu:187-195
            static const double u[7] = {0, 1, 2, 4, 2, 1, 0};
            for (int r = 0; r < 7; ++r)
                for (int c = 0; c < 7; ++c) {
                    double v = u[r] * u[c];
                    const int ar = std::abs(r - 3), ac = std::abs(c - 3);
                    if ((ar == 3 && ac == 0) || (ar == 0 && ac == 3)) v += 1.0;
                    if (ar == 2 && ac == 2) v -= 1.0;
                    p[r * 7 + c] = v;
                }
            return 49;
        }
        case LORA_BOX2D1R:
        case LORA_BOX2D3R: {
            // running counter over the upper-left octant, mirrored 8 ways; centre forced to 8
            Mat7 m;
            double num = 1.0;
            for (int i = -3; i <= 0; ++i)
                for (int j = i; j <= 0; ++j) {
                    set_symmetric8(m, i, j, num);
                    num += 1.0;
                }
            m.a[3][3] = 8.0;
            std::memcpy(p, m.a, sizeof(m.a));
            return 49;
        }
        case LORA_STAR2D3R: {
            Mat7 m;
            for (int k = 0; k < 7; ++k) {
                const double v = 4 - std::abs(k - 3);
                m.a[k][3] = v;
                m.a[3][k] = v;
            }
            std::memcpy(p, m.a, sizeof(m.a));
            return 49;
        }
        case LORA_STAR3D1R: {
            std::fill(p, p + 27, 0.0);
            p[13] = 2.0;
            p[13 - 9] = p[13 + 9] = 1.0;
            p[13 - 3] = p[13 + 3] = 1.0;
            p[13 - 1] = p[13 + 1] = 1.0;
            return 27;
        }
        case LORA_BOX3D1R: {
            static const double x[3] = {1, 2, 1};
            for (int k = 0; k < 27; ++k) p[k] = x[k % 3];
            return 27;
        }
        default:
            return LORA_EINVAL;
    }
}

// ---- low-rank factor precompute -------------------------------------------------------------
// Pyramid peeling: at level L the residual's support is the centred (7-2L)^2 square.  Its outer
// ring is matched by one rank-1 term whose row profile is the ring's first row and whose column
// profile is the ring's first column divided by the corner; rows are mirrored about the centre
// (the matrix must be symmetric under i -> -i for this to be exact).  Same arithmetic per entry
// as the reference (one divide per row, one multiply, one subtract), so the factors agree
// bit-for-bit with 2d/gpu.cu:280-350 for any params.
int factorize_7x7(const double *params, double u[4][7], double v[4][7], double *residual_max) {
    Mat7 res(params);
    std::memset(u, 0, sizeof(double) * 28);
    std::memset(v, 0, sizeof(double) * 28);

    for (int L = 0; L < 3; ++L) {
        const int lo = L, hi = 6 - L;
        const double corner = res.a[lo][lo];
        Mat7 term, next;
        // first ring row (and its mirror) is taken verbatim
        for (int c = lo; c <= hi; ++c) {
            term.a[lo][c] = res.a[lo][c];
            term.a[hi][c] = (L == 0) ? res.a[hi][c] : res.a[lo][c];
        }
        for (int r = lo + 1; r <= 3; ++r) {
            const double prop = res.a[r][lo] / corner;
            for (int c = lo; c <= hi; ++c) {
                const double t = prop * res.a[lo][c];
                term.a[r][c] = term.a[6 - r][c] = t;
                next.a[r][c] = next.a[6 - r][c] = res.a[r][c] - t;
            }
        }
        for (int k = lo; k <= hi; ++k) {
            u[L][k] = term.a[lo][k];
            v[L][k] = term.a[k][lo] / term.a[lo][lo];
        }
        res = next;
    }
    // what is left is the centre entry: a fourth, 1x1 term the reference computes and never uploads
    u[3][3] = 1.0;
    v[3][3] = res.a[3][3];

    if (residual_max) {
        double worst = 0.0;
        for (int r = 0; r < 7; ++r)
            for (int c = 0; c < 7; ++c) {
                double s = u[0][r] * v[0][c];
                s += u[1][r] * v[1][c];
                s += u[2][r] * v[2][c];
                const double d = std::fabs(params[r * 7 + c] - s);
                if (!(d <= worst)) worst = d;  // NaN-propagating max
            }
        *residual_max = worst;
    }
    return LORA_OK;
}

// ---- rank-revealing factorisation of an arbitrary 7x7 tap matrix ---------------------------
// One-sided Jacobi SVD, W = sum_k sigma_k a_k b_k^T with sigma_1 >= sigma_2 >= ...  The pyramid factoriser above
// only handles symmetric tables whose residuals vanish ring by ring; this one takes any W and tells how many terms it
// really needs: truncating after `rank` terms leaves a residual of spectral norm sigma_{rank+1} (Eckart-Young), so
// max|residual entry| <= sigma_{rank+1}.  Terms are returned as u_k = sigma_k a_k (vertical profile), v_k = b_k
// (horizontal profile); sigma[0..6] holds all singular values.
int svd_7x7(const double *W, double u[7][7], double v[7][7], double sigma[7]) {
    // work on G = W (rows i, columns j); rotate column pairs of G until they are mutually orthogonal: G = A S, V acc.
    double g[7][7], vv[7][7];
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) {
            g[i][j] = W[i * 7 + j];
            vv[i][j] = (i == j) ? 1.0 : 0.0;
            if (!std::isfinite(g[i][j])) return LORA_EINVAL;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 6; ++p)
            for (int q = p + 1; q < 7; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 7; ++i) {
                    alpha += g[i][p] * g[i][p];
                    beta += g[i][q] * g[i][q];
                    gamma += g[i][p] * g[i][q];
                }
                if (gamma == 0.0) continue;
                off = std::fmax(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 7; ++i) {
                    const double gp = g[i][p], gq = g[i][q];
                    g[i][p] = c * gp - sn * gq;
                    g[i][q] = sn * gp + c * gq;
                    const double vp = vv[i][p], vq = vv[i][q];
                    vv[i][p] = c * vp - sn * vq;
                    vv[i][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    // column norms are the singular values; sort descending
    int order[7];
    double nrm[7];
    for (int j = 0; j < 7; ++j) {
        double s2 = 0;
        for (int i = 0; i < 7; ++i) s2 += g[i][j] * g[i][j];
        nrm[j] = std::sqrt(s2);
        order[j] = j;
    }
    std::sort(order, order + 7, [&](int x, int y) { return nrm[x] > nrm[y]; });
    for (int k = 0; k < 7; ++k) {
        const int j = order[k];
        sigma[k] = nrm[j];
        for (int i = 0; i < 7; ++i) {
            u[k][i] = g[i][j];   // = sigma_k a_k
            v[k][i] = vv[i][j];  // = b_k
        }
    }
    return LORA_OK;
}

// Exact rank-1 test of a 3 x 3 x 3 fp32 tap tensor: w[dz][dy][dx] == (a[dz] * b[dy]) * c[dx] with every product
// rounded to fp32.  The factors are read off the three axes through the centre tap (c as stored, a and b divided
// by the centre), so a tensor that passes is evaluated with exactly these numbers (kernels_3d_bf16.hip, TAPS3D_SEP).
// cba = {c[0..2], b[0..2], a[0..2]}.  Returns 1 when the test holds.
int separable_27(const float *w, float *cba) {
    const float centre = w[13];
    if (!(centre != 0.0f) || !std::isfinite(centre)) return 0;
    float c[3], b[3], a[3];
    for (int i = 0; i < 3; ++i) {
        c[i] = w[9 + 3 + i];
        b[i] = w[9 + 3 * i + 1] / centre;
        a[i] = w[9 * i + 3 + 1] / centre;
    }
    for (int dz = 0; dz < 3; ++dz)
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const float ab = a[dz] * b[dy];
                const float abc = ab * c[dx];
                if (!(abc == w[dz * 9 + dy * 3 + dx])) return 0;
            }
    for (int i = 0; i < 3; ++i) {
        cba[i] = c[i];
        cba[3 + i] = b[i];
        cba[6 + i] = a[i];
    }
    return 1;
}

// The same test in fp64 (3D fp64 plane-streaming kernel, TAPS3D_SEP): w[dz][dy][dx] == (a[dz] * b[dy]) * c[dx] with every
// product rounded to fp64.  Holds for the reference's box taps -- they depend on dx only (3d/gpu_box.cu:151-164) -- and
// for any scaling of them.
int separable_27d(const double *w, double *cba) {
    const double centre = w[13];
    if (!(centre != 0.0) || !std::isfinite(centre)) return 0;
    double c[3], b[3], a[3];
    for (int i = 0; i < 3; ++i) {
        c[i] = w[9 + 3 + i];
        b[i] = w[9 + 3 * i + 1] / centre;
        a[i] = w[9 * i + 3 + 1] / centre;
    }
    for (int dz = 0; dz < 3; ++dz)
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const double ab = a[dz] * b[dy];
                const double abc = ab * c[dx];
                if (!(abc == w[dz * 9 + dy * 3 + dx])) return 0;
            }
    for (int i = 0; i < 3; ++i) {
        cba[i] = c[i];
        cba[3 + i] = b[i];
        cba[6 + i] = a[i];
    }
    return 1;
}

// Factors for the bf16 MFMA variant (kernels_3d_bf16_mfma.hip): separable taps as scale * a' (x) b' (x) c' with the
// normalised factors (divided by their first entry) EXACT in bf16, so that they can be matrix-instruction operands.
int mfma_factors_27(const float *cba, float *scale, float *cba_n) {
    auto bf16_exact = [](float x) {
        uint32_t u;
        std::memcpy(&u, &x, 4);
        return std::isfinite(x) && (u & 0xffffu) == 0;
    };
    for (int f = 0; f < 3; ++f) {
        const float first = cba[3 * f];
        if (!(first != 0.0f) || !std::isfinite(first)) return 0;
        for (int i = 0; i < 3; ++i) {
            const float q = cba[3 * f + i] / first;
            if (!bf16_exact(q) || !(q * first == cba[3 * f + i])) return 0;
            cba_n[3 * f + i] = q;
        }
    }
    const float ab = cba[6] * cba[3];  // a[0] b[0]
    *scale = ab * cba[0];
    return std::isfinite(*scale) && *scale != 0.0f;
}

int effective_weights(int shape, const double *params, double *w) {
    switch (shape) {
        case LORA_1D1R:
        case LORA_1D2R:
            std::copy(params, params + 9, w);
            return 9;
        case LORA_STAR2D1R:
            // the operator hard-codes its factors and correction; params are not read
            return default_params(LORA_STAR2D1R, w);
        case LORA_STAR2D3R: {
            std::fill(w, w + 49, 0.0);
            for (int k = 0; k < 7; ++k) {
                w[k * 7 + 3] = params[k * 7 + 3];             // vertical band: centre column
                if (k != 3) w[3 * 7 + k] = params[3 * 7 + k];  // horizontal band: centre row minus centre
            }
            return 49;
        }
        case LORA_BOX2D1R:
        case LORA_BOX2D3R: {
            double u[4][7], v[4][7];
            factorize_7x7(params, u, v, nullptr);
            // the kernel applies u_t down the rows (left band matrix) and v_t along the columns
            for (int r = 0; r < 7; ++r)
                for (int c = 0; c < 7; ++c) {
                    double s = u[0][r] * v[0][c];
                    s += u[1][r] * v[1][c];
                    s += u[2][r] * v[2][c];
                    w[r * 7 + c] = s;
                }
            return 49;
        }
        case LORA_STAR3D1R:
            return default_params(LORA_STAR3D1R, w);
        case LORA_BOX3D1R:
            for (int k = 0; k < 27; ++k) w[k] = params[k % 3];
            return 27;
        default:
            return LORA_EINVAL;
    }
}

}  // namespace lora

// ---- C ABI (group C of include/lorastencil.h) --------------------------------------------------
extern "C" {

int lora_shape_ntaps(int shape) { return lora::shape_ntaps(shape); }
int lora_shape_ndim(int shape) { return lora::shape_ndim(shape); }

int lora_shape_from_name(const char *name) {
    if (!name) return LORA_EINVAL;
    static const char *names[LORA_NUM_SHAPES] = {"1d1r",     "1d2r",    "star2d1r", "box2d1r",
                                                 "star2d3r", "box2d3r", "star3d1r", "box3d1r"};
    for (int s = 0; s < LORA_NUM_SHAPES; ++s)
        if (std::strcmp(name, names[s]) == 0) return s;
    return LORA_EINVAL;
}

const char *lora_shape_info_name(int shape) {
    // 1d/main.cu:6-9, 2d/main.cu:5-10, 3d/main.cu:5-8
    static const char *names[LORA_NUM_SHAPES] = {"1d1r",      "1d2r",     "star_2d1r", "box_2d1r",
                                                 "star_2d3r", "box_2d3r", "star_3d1r", "box_3d1r"};
    return (shape >= 0 && shape < LORA_NUM_SHAPES) ? names[shape] : "?";
}

size_t lora_padded_count(int shape, const int *dims) {
    if (!dims) return 0;
    switch (lora::shape_ndim(shape)) {
        case 1:
            return (size_t) dims[0] + 8;
        case 2:
            return ((size_t) dims[0] + 8) * ((size_t) dims[1] + 8);
        case 3:
            return ((size_t) dims[0] + 2) * ((size_t) dims[1] + 4) * ((size_t) dims[2] + 8);
        default:
            return 0;
    }
}

int lora_shape_gstencil_factor(int shape) {
    // 1d/gpu_1r.cu:132, 1d/gpu_2r.cu:134, 2d/gpu.cu:419, :478, :553, 3d/gpu_star.cu:190, 3d/gpu_box.cu:221
    switch (shape) {
        case LORA_1D1R:
        case LORA_STAR2D1R:
        case LORA_BOX2D1R:
        case LORA_BOX2D3R:
            return 3;
        case LORA_1D2R:
            return 2;
        case LORA_STAR2D3R:
        case LORA_STAR3D1R:
        case LORA_BOX3D1R:
            return 1;
        default:
            return 0;
    }
}

int lora_default_params(int shape, double *params) {
    if (!params) return LORA_EINVAL;
    return lora::default_params(shape, params);
}

int lora_effective_weights(int shape, const double *params, double *weights) {
    if (!weights) return LORA_EINVAL;
    double tmp[49];
    if (!params) {
        const int n = lora::default_params(shape, tmp);
        if (n < 0) return n;
        params = tmp;
    }
    return lora::effective_weights(shape, params, weights);
}

int lora_separable_3x3x3(const double *weights27, float *cba9) {
    if (!weights27 || !cba9) return LORA_EINVAL;
    float w[27];
    for (int k = 0; k < 27; ++k) w[k] = (float) weights27[k];
    return lora::separable_27(w, cba9);
}

int lora_svd_7x7(const double *weights, double *u, double *v, double *sigma) {
    if (!weights || !u || !v || !sigma) return LORA_EINVAL;
    return lora::svd_7x7(weights, reinterpret_cast<double(*)[7]>(u), reinterpret_cast<double(*)[7]>(v), sigma);
}

int lora_factorize_7x7(const double *params, double *u, double *v, double *residual_max) {
    if (!params || !u || !v) return LORA_EINVAL;
    return lora::factorize_7x7(params, reinterpret_cast<double(*)[7]>(u), reinterpret_cast<double(*)[7]>(v),
                               residual_max);
}

// host-side bf16 conversion (round-to-nearest-even; NaN stays a quiet NaN)
static uint16_t f32_to_bf16_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t) ((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t) (u >> 16);
}

void lora_f64_to_bf16(const double *src, uint16_t *dst, size_t count) {
    for (size_t i = 0; i < count; ++i) dst[i] = f32_to_bf16_bits((float) src[i]);
}

void lora_bf16_to_f64(const uint16_t *src, double *dst, size_t count) {
    for (size_t i = 0; i < count; ++i) {
        const uint32_t u = (uint32_t) src[i] << 16;
        float f;
        std::memcpy(&f, &u, 4);
        dst[i] = f;
    }
}

// glibc TYPE_3 generator: x[i] = x[i-3] + x[i-31] (mod 2^32), output x[i] >> 1; the 34-entry
// state is primed with a Lehmer sequence and the first 310 outputs are thrown away.
void lora_rng_seed(lora_rng *g, unsigned seed) {
    if (seed == 0) seed = 1;
    int64_t word = seed;
    g->r[0] = (int32_t) word;
    for (int i = 1; i < 31; ++i) {
        word = (16807 * word) % 2147483647;
        if (word < 0) word += 2147483647;
        g->r[i] = (int32_t) word;
    }
    for (int i = 31; i < 34; ++i) g->r[i] = g->r[i - 31];
    g->pos = 0;
    for (int i = 0; i < 310; ++i) (void) lora_rng_next(g);
}

int lora_rng_next(lora_rng *g) {
    const int p = g->pos;
    const int i3 = (p + 31 >= 34) ? p + 31 - 34 : p + 31;  // x[i-3]
    const int i31 = (p + 3 >= 34) ? p + 3 - 34 : p + 3;    // x[i-31]
    const uint32_t x = (uint32_t) g->r[i3] + (uint32_t) g->r[i31];
    g->r[p] = (int32_t) x;
    g->pos = (p + 1 == 34) ? 0 : p + 1;
    return (int) (x >> 1);
}

void lora_fill_rand(double *dst, size_t count, int mod, lora_rng *g) {
    for (size_t i = 0; i < count; ++i) dst[i] = (double) (lora_rng_next(g) % mod);
}

}  // extern "C"
