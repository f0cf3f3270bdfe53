/*
 * lorastencil_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see lorastencil_oracle.h).
 *
 * Plain C restatement of the reference's algorithm for the stencil hot path.  Every function
 * cites the reference file:line (under /root/reference/src/) it follows.  Compile with
 * -ffp-contract=off so that each tap is one multiply and one add, exactly like the
 * reference's test_cpu built for baseline x86-64 (no FMA contraction).
 */
#include "lorastencil_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * glibc rand(): TYPE_3 additive-feedback generator, r[i] = r[i-3] + r[i-31], output >> 1,
 * state seeded with the Lehmer sequence 16807*x mod (2^31-1) and 310 values discarded.
 * The reference never calls srand(), so the stream is that of seed 1 (1d/main.cu:107,
 * 2d/main.cu:234-236, 3d/main.cu:166-168).  Checked against this container's libc in
 * tests/test_oracle.py.
 * ---------------------------------------------------------------------------------------- */
void oracle_rng_seed(oracle_rng *g, unsigned seed) {
    int32_t init[34];
    if (seed == 0) seed = 1;
    init[0] = (int32_t) seed;
    for (int i = 1; i < 31; i++) {
        int64_t word = (16807LL * (int64_t) init[i - 1]) % 2147483647LL;
        if (word < 0) word += 2147483647LL;
        init[i] = (int32_t) word;
    }
    for (int i = 31; i < 34; i++) init[i] = init[i - 31];
    memcpy(g->r, init, sizeof(init));
    g->pos = 0;
    /* entries 34..343 are discarded */
    for (int i = 34; i < 344; i++) (void) oracle_rng_next(g);
}

int oracle_rng_next(oracle_rng *g) {
    /* ring of the last 34 values; pos = slot of r[i-34]; r[i-31] is 3 ahead, r[i-3] is 31 ahead */
    int p = g->pos;
    uint32_t a = (uint32_t) g->r[(p + 3) % 34];
    uint32_t b = (uint32_t) g->r[(p + 31) % 34];
    uint32_t v = a + b;
    g->r[p] = (int32_t) v;
    g->pos = (p + 1) % 34;
    return (int) (v >> 1);
}

void oracle_fill_rand(double *dst, size_t count, int mod, oracle_rng *g) {
    for (size_t i = 0; i < count; i++) dst[i] = (double) (oracle_rng_next(g) % mod);
}

/* ------------------------------------------------------------------------------------------
 * params tables of the reference harness
 * ---------------------------------------------------------------------------------------- */
static void default_box2d(double *p) {
    /* 2d/main.cu:149-167: eight-fold symmetric fill with a running counter, then centre := 8 */
    int num = 1;
    memset(p, 0, 49 * sizeof(double));
    for (int i = -3; i < 1; i++) {
        for (int j = -3; j < 1; j++) {
            if (i <= j) {
                const int a[2] = {i + 3, -i + 3};
                const int b[2] = {j + 3, -j + 3};
                for (int s = 0; s < 2; s++)
                    for (int t = 0; t < 2; t++) {
                        p[a[s] * 7 + b[t]] = num;
                        p[b[t] * 7 + a[s]] = num;
                    }
                num++;
            }
        }
    }
    p[3 * 7 + 3] = 8;
}

int oracle_default_params(int shape, double *p) {
    switch (shape) {
        case ORACLE_1D1R: { /* 1d/main.cu:77 */
            static const double w[9] = {0, 1, 2, 3, 4, 3, 2, 1, 0};
            memcpy(p, w, sizeof(w));
            return 9;
        }
        case ORACLE_1D2R: { /* 1d/main.cu:78 */
            static const double w[9] = {1, 2, 3, 4, 5, 4, 3, 2, 1};
            memcpy(p, w, sizeof(w));
            return 9;
        }
        case ORACLE_STAR2D1R: { /* 2d/main.cu:187-195 */
            static const double w[49] = {0, 0, 0, 1,  0, 0, 0, 0, 0, 2, 4, 2, 0, 0, 0, 2, 4, 8, 4, 2, 0, 1, 4, 8, 16,
                                         8, 4, 1, 0,  2, 4, 8, 4, 2, 0, 0, 0, 2, 4, 2, 0, 0, 0, 0, 0, 1, 0, 0, 0};
            memcpy(p, w, sizeof(w));
            return 49;
        }
        case ORACLE_BOX2D1R:
        case ORACLE_BOX2D3R: /* 2d/main.cu:200-211: both box shapes use param_box_2d1r */
            default_box2d(p);
            return 49;
        case ORACLE_STAR2D3R: { /* 2d/main.cu:176-184 */
            int num = 1;
            memset(p, 0, 49 * sizeof(double));
            for (int i = -3; i < 1; i++) {
                p[(i + 3) * 7 + 3] = num;
                p[(-i + 3) * 7 + 3] = num;
                p[3 * 7 + (i + 3)] = num;
                p[3 * 7 + (-i + 3)] = num;
                num++;
            }
            return 49;
        }
        case ORACLE_STAR3D1R: { /* 3d/main.cu:121-125 */
            static const double w[27] = {0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 1, 2, 1, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0};
            memcpy(p, w, sizeof(w));
            return 27;
        }
        case ORACLE_BOX3D1R: /* 3d/main.cu:112-119 */
            p[0] = 1;
            p[1] = 2;
            p[2] = 1;
            for (int i = 3; i < 27; i++) p[i] = p[i % 3];
            return 27;
        default:
            return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * Pyramid factorisation (2d/gpu.cu:280-350).  Level L peels the outermost ring of the
 * (7-2L)x(7-2L) residual: rows are assumed proportional to the ring's first row, the
 * proportionality factor is read off the ring's first column, rows are mirrored i -> -i.
 * Arithmetic (one divide per row, one multiply and one subtract per entry) is kept in the
 * reference's order so that the factors are bit-identical for any params.
 * ---------------------------------------------------------------------------------------- */
void oracle_factorize_7x7(const double *params, double u[4][7], double v[4][7]) {
    double fact[4][49];
    double resid[3][49]; /* resid[L] = residual after removing term L */
    const double *src = params;
    memset(fact, 0, sizeof(fact));
    memset(resid, 0, sizeof(resid));

    for (int L = 0; L < 3; L++) {
        const int lo = L, hi = 6 - L; /* ring spans rows/cols lo..hi */
        for (int r = lo; r <= 3; r++) {
            const int mr = 6 - r; /* mirrored row */
            if (r == lo) {
                for (int c = lo; c <= hi; c++) {
                    /* 2d/gpu.cu:285-286 copies both rows from params at level 0; levels 1 and 2
                     * (:304-305, :321-322) copy the top row into both */
                    fact[L][r * 7 + c] = src[r * 7 + c];
                    fact[L][mr * 7 + c] = (L == 0) ? src[mr * 7 + c] : src[r * 7 + c];
                }
            } else {
                const double prop = src[r * 7 + lo] / src[lo * 7 + lo];
                for (int c = lo; c <= hi; c++) {
                    const double f = prop * src[lo * 7 + c];
                    fact[L][r * 7 + c] = f;
                    fact[L][mr * 7 + c] = f;
                    const double d = src[r * 7 + c] - f;
                    if (L < 2) {
                        resid[L][r * 7 + c] = d;
                        resid[L][mr * 7 + c] = d;
                    } else {
                        fact[3][r * 7 + c] = d; /* 2d/gpu.cu:329-330: only row 3 reaches here */
                    }
                }
            }
        }
        if (L < 2) src = resid[L];
    }

    memset(u, 0, 4 * 7 * sizeof(double));
    memset(v, 0, 4 * 7 * sizeof(double));
    for (int L = 0; L < 3; L++) { /* 2d/gpu.cu:337-348 */
        for (int i = L; i <= 6 - L; i++) {
            u[L][i] = fact[L][L * 7 + i];
            v[L][i] = fact[L][i * 7 + L] / fact[L][L * 7 + L];
        }
    }
    u[3][3] = 1.0; /* 2d/gpu.cu:349-350 */
    v[3][3] = fact[3][3 * 7 + 3];
}

int oracle_effective_weights(int shape, const double *params, double *w) {
    switch (shape) {
        case ORACLE_1D1R:
        case ORACLE_1D2R: /* P[(row+col)*8+col] = params[row] (1d/gpu_1r.cu:95-99) */
            memcpy(w, params, 9 * sizeof(double));
            return 9;
        case ORACLE_STAR2D1R: { /* params ignored (2d/gpu.cu:486-487) */
            static const double uv[7] = {0, 1, 2, 4, 2, 1, 0};
            for (int dy = 0; dy < 7; dy++)
                for (int dx = 0; dx < 7; dx++) w[dy * 7 + dx] = uv[dy] * uv[dx];
            /* 8-point correction from shared memory (2d/gpu.cu:254-262) */
            w[3 * 7 + 0] += 1.0;
            w[3 * 7 + 6] += 1.0;
            w[0 * 7 + 3] += 1.0;
            w[6 * 7 + 3] += 1.0;
            w[1 * 7 + 1] -= 1.0;
            w[1 * 7 + 5] -= 1.0;
            w[5 * 7 + 1] -= 1.0;
            w[5 * 7 + 5] -= 1.0;
            return 49;
        }
        case ORACLE_STAR2D3R: /* 2d/gpu.cu:433-444 */
            memset(w, 0, 49 * sizeof(double));
            for (int k = 0; k < 7; k++) {
                if (k != 3) w[3 * 7 + k] = params[3 * 7 + k]; /* V: centre row, centre tap removed */
            }
            for (int k = 0; k < 7; k++) w[k * 7 + 3] = params[k * 7 + 3]; /* U: centre column */
            return 49;
        case ORACLE_BOX2D1R:
        case ORACLE_BOX2D3R: { /* only terms 0..2 are uploaded (2d/gpu.cu:358-369, :389-390) */
            double u[4][7], v[4][7];
            oracle_factorize_7x7(params, u, v);
            /* U band multiplies from the left (vertical), V from the right (horizontal):
             * out = sum_t (U_t X) V_t (2d/gpu.cu:68-101) */
            for (int dy = 0; dy < 7; dy++)
                for (int dx = 0; dx < 7; dx++) {
                    double s = u[0][dy] * v[0][dx];
                    s += u[1][dy] * v[1][dx];
                    s += u[2][dy] * v[2][dx];
                    w[dy * 7 + dx] = s;
                }
            return 49;
        }
        case ORACLE_STAR3D1R: { /* params ignored (3d/gpu_star.cu:51, :142-151) */
            memset(w, 0, 27 * sizeof(double));
            w[0 * 9 + 1 * 3 + 1] = 1;
            w[2 * 9 + 1 * 3 + 1] = 1;
            w[1 * 9 + 0 * 3 + 1] = 1;
            w[1 * 9 + 2 * 3 + 1] = 1;
            w[1 * 9 + 1 * 3 + 0] = 1;
            w[1 * 9 + 1 * 3 + 2] = 1;
            w[1 * 9 + 1 * 3 + 1] = 2;
            return 27;
        }
        case ORACLE_BOX3D1R: /* ones in z and y, params[0..2] along x (3d/gpu_box.cu:151-164, :127-138) */
            for (int dz = 0; dz < 3; dz++)
                for (int dy = 0; dy < 3; dy++)
                    for (int dx = 0; dx < 3; dx++) w[dz * 9 + dy * 3 + dx] = params[dx];
            return 27;
        default:
            return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * One kernel application, left-to-right over the taps like the reference's test_cpu.
 * ---------------------------------------------------------------------------------------- */
static int pick_threads(int threads) {
#ifdef _OPENMP
    if (threads <= 0) return omp_get_max_threads();
    return threads;
#else
    (void) threads;
    return 1;
#endif
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_step_1d(const double *in, double *out, const double *w, int cols, int threads) {
    /* 1d/main.cu:34-40; the radius is 4 in both 1D shapes */
    const int nt = pick_threads(threads);
    (void) nt;
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
    for (int col = 4; col < cols - 4; col++) {
        double s = w[0] * in[col - 4];
        for (int t = 1; t < 9; t++) s = s + w[t] * in[col - 4 + t];
        out[col] = s;
    }
}

void oracle_step_2d(const double *in, double *out, const double *w, int rows, int cols, int threads) {
    /* 2d/main.cu:38-93: rows/cols are the PADDED extents, halo 4, taps in row-major order */
    const int nt = pick_threads(threads);
    (void) nt;
    ptrdiff_t off[49]; /* tap k = (dy, dx) in row-major order, as the reference writes the sum out */
    for (int k = 0; k < 49; k++) off[k] = (k / 7 - 3) * (ptrdiff_t) cols + (k % 7 - 3);
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
    for (int row = 4; row < rows - 4; row++) {
        for (int col = 4; col < cols - 4; col++) {
            const double *c = in + (size_t) row * cols + col;
            double s = w[0] * c[off[0]];
            for (int k = 1; k < 49; k++) s = s + w[k] * c[off[k]];
            out[(size_t) row * cols + col] = s;
        }
    }
}

void oracle_step_3d(const double *in, double *out, const double *w, int heights, int rows, int cols, int threads) {
    /* 3d/main.cu:33-68: halos z 1, y 2, x 4 (3d/main.cu:21-23) */
    const int nt = pick_threads(threads);
    (void) nt;
    const ptrdiff_t plane = (ptrdiff_t) rows * cols;
    ptrdiff_t off[27]; /* tap k = (dz, dy, dx) in row-major order */
    for (int k = 0; k < 27; k++) off[k] = (k / 9 - 1) * plane + ((k / 3) % 3 - 1) * (ptrdiff_t) cols + (k % 3 - 1);
#pragma omp parallel for num_threads(nt) schedule(static) collapse(2) if (nt > 1)
    for (int h = 1; h < heights - 1; h++) {
        for (int row = 2; row < rows - 2; row++) {
            for (int col = 4; col < cols - 4; col++) {
                const double *c = in + h * plane + (ptrdiff_t) row * cols + col;
                double s = w[0] * c[off[0]];
                for (int k = 1; k < 27; k++) s = s + w[k] * c[off[k]];
                out[h * plane + (ptrdiff_t) row * cols + col] = s;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Operator = reference driver semantics
 * ---------------------------------------------------------------------------------------- */
static int shape_dim(int shape) {
    if (shape == ORACLE_1D1R || shape == ORACLE_1D2R) return 1;
    if (shape >= ORACLE_STAR2D1R && shape <= ORACLE_BOX2D3R) return 2;
    if (shape == ORACLE_STAR3D1R || shape == ORACLE_BOX3D1R) return 3;
    return 0;
}

size_t oracle_padded_count(int shape, const int *dims) {
    switch (shape_dim(shape)) {
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

int oracle_run_weights(int shape, const double *in, double *out, const double *w, int times, const int *dims,
                       int threads) {
    const int nd = shape_dim(shape);
    if (nd == 0 || times < 0) return -1;
    const size_t count = oracle_padded_count(shape, dims);
    double *buf[2];
    buf[0] = (double *) malloc(count * sizeof(double));
    buf[1] = (double *) calloc(count, sizeof(double)); /* cudaMemset(array_d[1], 0) 2d/gpu.cu:533 */
    if (!buf[0] || !buf[1]) {
        free(buf[0]);
        free(buf[1]);
        return -1;
    }
    memcpy(buf[0], in, count * sizeof(double)); /* whole padded input, halo included (2d/gpu.cu:532) */
    for (int i = 0; i < times; i++) {           /* 2d/gpu.cu:544-546 */
        const double *src = buf[i % 2];
        double *dst = buf[(i + 1) % 2];
        if (nd == 1)
            oracle_step_1d(src, dst, w, dims[0] + 8, threads);
        else if (nd == 2)
            oracle_step_2d(src, dst, w, dims[0] + 8, dims[1] + 8, threads);
        else
            oracle_step_3d(src, dst, w, dims[0] + 2, dims[1] + 4, dims[2] + 8, threads);
    }
    /* 1D copies array_size - sizeof(double) (1d/gpu_1r.cu:134): the last element of out is untouched */
    const size_t ncopy = (nd == 1) ? count - 1 : count;
    memcpy(out, buf[times % 2], ncopy * sizeof(double)); /* 2d/gpu.cu:554 */
    free(buf[0]);
    free(buf[1]);
    return 0;
}

int oracle_run(int shape, const double *in, double *out, const double *params, int times, const int *dims,
               int threads) {
    double w[49];
    if (oracle_effective_weights(shape, params, w) < 0) return -1;
    return oracle_run_weights(shape, in, out, w, times, dims, threads);
}

/* ------------------------------------------------------------------------------------------
 * bf16 storage, fp32 accumulation (NEW capability; parity unpinned by the reference, see header)
 * ---------------------------------------------------------------------------------------- */
uint16_t oracle_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t) ((u >> 16) | 0x0040u); /* NaN stays a (quiet) NaN */
    u += 0x7fffu + ((u >> 16) & 1u);                                               /* round to nearest even */
    return (uint16_t) (u >> 16);
}

float oracle_bf16_to_f32(uint16_t b) {
    const uint32_t u = (uint32_t) b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void oracle_step_3d_bf16(const uint16_t *in, uint16_t *out, const float *w, int heights, int rows, int cols,
                         int threads) {
    const int nt = pick_threads(threads);
    (void) nt;
    const ptrdiff_t plane = (ptrdiff_t) rows * cols;
    ptrdiff_t off[27];
    for (int k = 0; k < 27; k++) off[k] = (k / 9 - 1) * plane + ((k / 3) % 3 - 1) * (ptrdiff_t) cols + (k % 3 - 1);
#pragma omp parallel for num_threads(nt) schedule(static) collapse(2) if (nt > 1)
    for (int h = 1; h < heights - 1; h++) {
        for (int row = 2; row < rows - 2; row++) {
            for (int col = 4; col < cols - 4; col++) {
                const uint16_t *c = in + h * plane + (ptrdiff_t) row * cols + col;
                float s = 0.0f;
                for (int k = 0; k < 27; k++) s = fmaf(w[k], oracle_bf16_to_f32(c[off[k]]), s);
                out[h * plane + (ptrdiff_t) row * cols + col] = oracle_f32_to_bf16(s);
            }
        }
    }
}

/* Exact rank-1 test of fp32 3x3x3 taps (the engine's lora_separable_3x3x3 restated): factors read off the three
 * axes through the centre tap; holds iff fl(fl(a[dz] * b[dy]) * c[dx]) == w[dz][dy][dx] for all 27 taps. */
int oracle_separable_27(const float *w, float *c, float *b, float *a) {
    const float centre = w[13];
    if (!(centre != 0.0f) || isinf(centre)) return 0;
    for (int i = 0; i < 3; i++) {
        c[i] = w[12 + i];
        b[i] = w[9 + 3 * i + 1] / centre;
        a[i] = w[9 * i + 4] / centre;
    }
    for (int k = 0; k < 27; k++) {
        const float ab = a[k / 9] * b[(k / 3) % 3];
        const float abc = ab * c[k % 3];
        if (!(abc == w[k])) return 0;
    }
    return 1;
}

/* One bf16 sweep of separable taps in the engine's documented order (kernels_3d_bf16.hip, TAPS3D_SEP):
 *   T = fma(c2, x[+1], fma(c1, x[0], c0 * x[-1]))      along x, fp32
 *   U = fma(b2, T[+1], fma(b1, T[0], b0 * T[-1]))      along y
 *   out = bf16(fma(a2, U[+1], fma(a1, U[0], a0 * U[-1])))   along z, one rounding to bf16
 * (built with -ffp-contract=off, so the plain products stay separate roundings). */
void oracle_step_3d_bf16_sep(const uint16_t *in, uint16_t *out, const float *c, const float *b, const float *a,
                             int heights, int rows, int cols, int threads) {
    const int nt = pick_threads(threads);
    (void) nt;
    const ptrdiff_t plane = (ptrdiff_t) rows * cols;
#pragma omp parallel for num_threads(nt) schedule(static) collapse(2) if (nt > 1)
    for (int h = 1; h < heights - 1; h++) {
        for (int row = 2; row < rows - 2; row++) {
            for (int col = 4; col < cols - 4; col++) {
                float U[3];
                for (int dz = 0; dz < 3; dz++) {
                    float T[3];
                    for (int dy = 0; dy < 3; dy++) {
                        const uint16_t *x = in + (h + dz - 1) * plane + (ptrdiff_t) (row + dy - 1) * cols + col;
                        float t = c[0] * oracle_bf16_to_f32(x[-1]);
                        t = fmaf(c[1], oracle_bf16_to_f32(x[0]), t);
                        t = fmaf(c[2], oracle_bf16_to_f32(x[1]), t);
                        T[dy] = t;
                    }
                    float u = b[0] * T[0];
                    u = fmaf(b[1], T[1], u);
                    u = fmaf(b[2], T[2], u);
                    U[dz] = u;
                }
                float o = a[0] * U[0];
                o = fmaf(a[1], U[1], o);
                o = fmaf(a[2], U[2], o);
                out[h * plane + (ptrdiff_t) row * cols + col] = oracle_f32_to_bf16(o);
            }
        }
    }
}

/* Contract of the engine's bf16 MATRIX-PIPE variant (kernels_3d_bf16_mfma.hip; no reference counterpart, unpinned):
 * taps = scale * a' (x) b' (x) c' with the normalised factors (each divided by its first entry) exact in bf16;
 *   S   = the 27-term sum  a'[dz] b'[dy] c'[dx] x  -- the matrix instruction accumulates it in fp32, which is EXACT
 *         while the addends of a point span fewer than ~14 binary orders of magnitude; restated here as the exact sum
 *         (double: 27 products of 8-bit by <= 8-bit significands) rounded once to fp32
 *   out = bf16( fl32( scale * S ) ),  one round-to-nearest-even to bf16 per sweep. */
int oracle_mfma_factors(const float *c, const float *b, const float *a, float *scale, float *cn, float *bn, float *an) {
    const float *f[3] = {c, b, a};
    float *o[3] = {cn, bn, an};
    for (int k = 0; k < 3; k++) {
        const float first = f[k][0];
        if (!(first != 0.0f) || !isfinite(first)) return 0;
        for (int i = 0; i < 3; i++) {
            const float q = f[k][i] / first;
            uint32_t u;
            memcpy(&u, &q, 4);
            if (!isfinite(q) || (u & 0xffffu) != 0 || !(q * first == f[k][i])) return 0;
            o[k][i] = q;
        }
    }
    const float ab = a[0] * b[0];
    *scale = ab * c[0];
    return isfinite(*scale) && *scale != 0.0f;
}

void oracle_step_3d_bf16_mfma(const uint16_t *in, uint16_t *out, float scale, const float *c, const float *b,
                              const float *a, int heights, int rows, int cols, int threads) {
    const int nt = pick_threads(threads);
    (void) nt;
    const ptrdiff_t plane = (ptrdiff_t) rows * cols;
#pragma omp parallel for num_threads(nt) schedule(static) collapse(2) if (nt > 1)
    for (int h = 1; h < heights - 1; h++) {
        for (int row = 2; row < rows - 2; row++) {
            for (int col = 4; col < cols - 4; col++) {
                double S = 0.0;
                for (int dz = 0; dz < 3; dz++)
                    for (int dy = 0; dy < 3; dy++)
                        for (int dx = 0; dx < 3; dx++) {
                            const uint16_t *x = in + (h + dz - 1) * plane + (ptrdiff_t) (row + dy - 1) * cols + col + dx - 1;
                            S += (double) a[dz] * (double) b[dy] * (double) c[dx] * (double) oracle_bf16_to_f32(*x);
                        }
                const float s32 = (float) S;
                const float o = scale * s32;
                out[h * plane + (ptrdiff_t) row * cols + col] = oracle_f32_to_bf16(o);
            }
        }
    }
}

int oracle_run_bf16(int shape, const uint16_t *in, uint16_t *out, const double *w27, int times, const int *dims,
                    int threads) {
    return oracle_run_bf16_mode(shape, in, out, w27, times, dims, threads, 0);
}

/* separable != 0: the engine's default -- exactly separable taps go through oracle_step_3d_bf16_sep. */
int oracle_run_bf16_mode(int shape, const uint16_t *in, uint16_t *out, const double *w27, int times, const int *dims,
                         int threads, int separable) {
    if (shape_dim(shape) != 3 || times < 0) return -1;
    const size_t count = oracle_padded_count(shape, dims);
    float w[27];
    for (int k = 0; k < 27; k++) w[k] = (float) w27[k];
    uint16_t *buf[2];
    buf[0] = (uint16_t *) malloc(count * sizeof(uint16_t));
    buf[1] = (uint16_t *) calloc(count, sizeof(uint16_t));
    if (!buf[0] || !buf[1]) {
        free(buf[0]);
        free(buf[1]);
        return -1;
    }
    memcpy(buf[0], in, count * sizeof(uint16_t));
    float c[3], b[3], a[3];
    const int sep = separable && oracle_separable_27(w, c, b, a);
    float scale = 0.0f, cn[3], bn[3], an[3];
    if (separable == 2 && !(sep && oracle_mfma_factors(c, b, a, &scale, cn, bn, an))) {
        free(buf[0]);
        free(buf[1]);
        return -1;  /* these taps have no matrix-pipe form */
    }
    for (int i = 0; i < times; i++) {
        if (separable == 2)
            oracle_step_3d_bf16_mfma(buf[i % 2], buf[(i + 1) % 2], scale, cn, bn, an, dims[0] + 2, dims[1] + 4, dims[2] + 8,
                                     threads);
        else if (sep)
            oracle_step_3d_bf16_sep(buf[i % 2], buf[(i + 1) % 2], c, b, a, dims[0] + 2, dims[1] + 4, dims[2] + 8, threads);
        else
            oracle_step_3d_bf16(buf[i % 2], buf[(i + 1) % 2], w, dims[0] + 2, dims[1] + 4, dims[2] + 8, threads);
    }
    memcpy(out, buf[times % 2], count * sizeof(uint16_t));
    free(buf[0]);
    free(buf[1]);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Boundary-condition options (not reference behaviour; see header)
 * ---------------------------------------------------------------------------------------- */
static void shape_geometry(int shape, const int *dims, int *nd, long ext[3], long hw[3]) {
    static const long h1[1] = {4}, h2[2] = {4, 4}, h3[3] = {1, 2, 4};
    *nd = shape_dim(shape);
    const long *h = *nd == 1 ? h1 : (*nd == 2 ? h2 : h3);
    for (int d = 0; d < 3; d++) {
        const int sd = d - (3 - *nd);
        ext[d] = sd >= 0 ? dims[sd] : 1;
        hw[d] = sd >= 0 ? h[sd] : 0;
    }
}

/* mode 0: dst halo <- src halo; mode 2: dst halo <- periodic image inside dst */
static void halo_apply(double *dst, const double *src, const long ext[3], const long hw[3], int wrap) {
    const long P0 = ext[0] + 2 * hw[0], P1 = ext[1] + 2 * hw[1], P2 = ext[2] + 2 * hw[2];
    for (long a = 0; a < P0; a++)
        for (long b = 0; b < P1; b++)
            for (long c = 0; c < P2; c++) {
                const int inside = a >= hw[0] && a < hw[0] + ext[0] && b >= hw[1] && b < hw[1] + ext[1] &&
                                   c >= hw[2] && c < hw[2] + ext[2];
                if (inside) continue;
                const long off = (a * P1 + b) * P2 + c;
                if (!wrap) {
                    dst[off] = src[off];
                } else {
                    long w[3] = {a, b, c};
                    for (int d = 0; d < 3; d++) {
                        if (w[d] < hw[d])
                            w[d] += ext[d];
                        else if (w[d] >= hw[d] + ext[d])
                            w[d] -= ext[d];
                    }
                    dst[off] = dst[(w[0] * P1 + w[1]) * P2 + w[2]];
                }
            }
}

int oracle_run_bc(int shape, const double *in, double *out, const double *w, int times, const int *dims, int bc,
                  int threads) {
    if (bc == ORACLE_BC_REFERENCE) return oracle_run_weights(shape, in, out, w, times, dims, threads);
    int nd;
    long ext[3], hw[3];
    shape_geometry(shape, dims, &nd, ext, hw);
    if (nd == 0 || times < 0) return -1;
    const size_t count = oracle_padded_count(shape, dims);
    double *buf[2];
    buf[0] = (double *) malloc(count * sizeof(double));
    buf[1] = (double *) calloc(count, sizeof(double));
    if (!buf[0] || !buf[1]) {
        free(buf[0]);
        free(buf[1]);
        return -1;
    }
    memcpy(buf[0], in, count * sizeof(double));
    if (bc == ORACLE_BC_DIRICHLET && times > 0) halo_apply(buf[1], buf[0], ext, hw, 0);
    for (int i = 0; i < times; i++) {
        double *src = buf[i % 2], *dst = buf[(i + 1) % 2];
        if (bc == ORACLE_BC_PERIODIC) halo_apply(src, NULL, ext, hw, 1);
        if (nd == 1)
            oracle_step_1d(src, dst, w, dims[0] + 8, threads);
        else if (nd == 2)
            oracle_step_2d(src, dst, w, dims[0] + 8, dims[1] + 8, threads);
        else
            oracle_step_3d(src, dst, w, dims[0] + 2, dims[1] + 4, dims[2] + 8, threads);
    }
    if (bc == ORACLE_BC_PERIODIC && times > 0) halo_apply(buf[times % 2], NULL, ext, hw, 1);
    memcpy(out, buf[times % 2], count * sizeof(double));
    free(buf[0]);
    free(buf[1]);
    return 0;
}
