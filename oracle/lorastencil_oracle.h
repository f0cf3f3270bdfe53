/*
 * lorastencil_oracle.h -- CPU oracle for the LoRAStencil hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of what the reference
 * (zondie17/LoRAStencil) computes behind `lorastencil_{1d,2d,3d} shape sizes times`.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product (lorastencil_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity pin: the one-step arithmetic is checked bit-for-bit against the reference's
 * own `test_cpu` functions compiled from /root/reference (oracle/_ref, see
 * oracle/build_ref.sh) and against the fixtures generated from them (tests/golden/).
 * Multi-step behaviour is NOT pinned by the reference (it has no multi-step check):
 * it follows the driver semantics read from the reference host code cited below.
 *
 * All file:line citations are into /root/reference/src/.
 */
#ifndef LORASTENCIL_ORACLE_H
#define LORASTENCIL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shape ids (shared numbering with include/lorastencil.h; restated, not included,
 * so that the oracle stays independent of the product headers). */
enum {
    ORACLE_1D1R = 0,      /* 1d/main.cu:53  */
    ORACLE_1D2R = 1,      /* 1d/main.cu:55  */
    ORACLE_STAR2D1R = 2,  /* 2d/main.cu:108 */
    ORACLE_BOX2D1R = 3,   /* 2d/main.cu:106 (same operator + weights as box2d3r, 2d/main.cu:276-279) */
    ORACLE_STAR2D3R = 4,  /* 2d/main.cu:110 */
    ORACLE_BOX2D3R = 5,   /* 2d/main.cu:112 */
    ORACLE_STAR3D1R = 6,  /* 3d/main.cu:83  */
    ORACLE_BOX3D1R = 7    /* 3d/main.cu:81  */
};

/* ---- glibc rand() (TYPE_3 additive feedback, default seed 1) ------------------- */
typedef struct {
    int32_t r[34];
    int pos; /* index of the oldest entry */
} oracle_rng;

void oracle_rng_seed(oracle_rng *g, unsigned seed);
int oracle_rng_next(oracle_rng *g); /* == glibc rand() */

/* Fill `count` doubles with (double)(rand() % mod), continuing the stream in `g`.
 * 1D: mod 10000 over n+9 draws (1d/main.cu:105-109); 2D/3D: mod 100 over the whole
 * padded array (2d/main.cu:232-236, 3d/main.cu:164-168). */
void oracle_fill_rand(double *dst, size_t count, int mod, oracle_rng *g);

/* ---- weights the reference harness passes as `params` --------------------------- */
/* Writes 9 / 49 / 27 doubles; returns the count (1d/main.cu:77-78, 2d/main.cu:139-195,
 * 3d/main.cu:112-125). */
int oracle_default_params(int shape, double *params);

/* Weights the reference GPU operator ACTUALLY applies for `params`:
 *  - 1d: params as given                               (1d/gpu_1r.cu:92-99)
 *  - star2d1r: hard-coded u (x) u + 8-point correction (2d/gpu.cu:486-487, :249-264)
 *  - star2d3r: centre column + centre row of params    (2d/gpu.cu:430-444)
 *  - box2d*:  sum of the first three pyramid terms     (2d/gpu.cu:280-369)
 *  - star3d1r: hard-coded 7-point                      (3d/gpu_star.cu:51, :142-151)
 *  - box3d1r: params[0..2] along x, ones in y and z    (3d/gpu_box.cu:151-164)
 * Writes 9 / 49 / 27 doubles; returns the count. */
int oracle_effective_weights(int shape, const double *params, double *w);

/* The reference's pyramid factorisation of a symmetric 7x7 matrix (2d/gpu.cu:280-350):
 * u[t][7], v[t][7] for t = 0..3 (term 3 is computed but never used by the kernel). */
void oracle_factorize_7x7(const double *params, double u[4][7], double v[4][7]);

/* ---- one kernel application, same summation order as the reference's test_cpu ---- */
/* `threads` > 1 uses OpenMP over rows/planes (each output point is still one
 * left-to-right sum, so results do not depend on the thread count). */
void oracle_step_1d(const double *in, double *out, const double *w9, int cols, int threads);               /* 1d/main.cu:34-40 */
void oracle_step_2d(const double *in, double *out, const double *w49, int rows, int cols, int threads);    /* 2d/main.cu:38-93 */
void oracle_step_3d(const double *in, double *out, const double *w27, int heights, int rows, int cols,
                    int threads);                                                                         /* 3d/main.cu:33-68 */

/* ---- the operator: reference driver semantics ---------------------- */
/* buf0 <- whole padded input, buf1 <- 0; step i reads buf[i%2], writes the interior of
 * buf[(i+1)%2]; out <- buf[times%2] (whole padded array; 1D: all but the last element)
 * (1d/gpu_1r.cu:103-134, 2d/gpu.cu:525-554, 3d/gpu_star.cu:158-192).
 * dims: 1D {n}, 2D {m, n}, 3D {h, m, n} (interior sizes). Returns 0, or -1 on bad args. */
int oracle_run(int shape, const double *in, double *out, const double *params, int times, const int *dims,
               int threads);

/* Same but with explicit weights (9/49/27) instead of params -> effective weights; used for the
 * normalised-weights mode and for multi-GPU slab tests. */
int oracle_run_weights(int shape, const double *in, double *out, const double *w, int times, const int *dims,
                       int threads);

/* ---- boundary-condition options of the driver (NOT behaviour of the reference, whose only
 * boundary condition is ORACLE_BC_REFERENCE = oracle_run_weights; parity for the other two is unpinned by it).
 *   DIRICHLET: halo cells hold the caller's input values at every time level (buffer 1 starts with buffer 0's halo);
 *   PERIODIC : before every sweep, and once after the last, each halo cell of the current buffer takes the value of
 *              the interior cell it is a periodic image of. */
enum { ORACLE_BC_REFERENCE = 0, ORACLE_BC_DIRICHLET = 1, ORACLE_BC_PERIODIC = 2 };
int oracle_run_bc(int shape, const double *in, double *out, const double *w, int times, const int *dims, int bc,
                  int threads);

/* ---- bf16 storage (3D shapes).  PARITY UNPINNED: the reference has no reduced-precision path at all
 * (SURVEY section 2 row 12), so there is nothing of its own to pin this against.  The contract restated here is the
 * one the engine documents (kernels_3d_bf16.hip): bf16 values, fp32 taps, one fp32 fused multiply-add per tap in
 * test_cpu's tap order (3d/main.cu:33-68) starting from 0, one round-to-nearest-even to bf16 per sweep; driver
 * semantics as in oracle_run. */
uint16_t oracle_f32_to_bf16(float f);
float oracle_bf16_to_f32(uint16_t b);
void oracle_step_3d_bf16(const uint16_t *in, uint16_t *out, const float *w27, int heights, int rows, int cols,
                         int threads);
int oracle_run_bf16(int shape, const uint16_t *in, uint16_t *out, const double *w27, int times, const int *dims,
                    int threads);
/* Separable taps (w = a (x) b (x) c exactly in fp32; every box3d1r of the reference's API): the engine's bf16 plans
 * evaluate x-, y-, z-passes; the order is restated in oracle_step_3d_bf16_sep.  oracle_run_bf16_mode(..., 1) is the
 * engine's default behaviour (separable form when the exact test holds), (..., 0) the 27-tap order. */
int oracle_separable_27(const float *w27, float *c, float *b, float *a);
void oracle_step_3d_bf16_sep(const uint16_t *in, uint16_t *out, const float *c, const float *b, const float *a,
                             int heights, int rows, int cols, int threads);
/* separable == 2: the contract of the engine's matrix-pipe variant (oracle_step_3d_bf16_mfma in the .c file):
 * exact 27-term sum with bf16-exact normalised factors, rounded to fp32, scaled once, rounded to bf16. */
void oracle_step_3d_bf16_mfma(const uint16_t *in, uint16_t *out, float scale, const float *c, const float *b,
                              const float *a, int heights, int rows, int cols, int threads);
int oracle_mfma_factors(const float *c, const float *b, const float *a, float *scale, float *cn, float *bn, float *an);
int oracle_run_bf16_mode(int shape, const uint16_t *in, uint16_t *out, const double *w27, int times, const int *dims,
                         int threads, int separable);

/* Padded element count of a shape/dims (1D n+8; 2D (m+8)(n+8); 3D (h+2)(m+4)(n+8)). */
size_t oracle_padded_count(int shape, const int *dims);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
