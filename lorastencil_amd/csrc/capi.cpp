// capi.cpp -- implementation of the C ABI in include/lorastencil.h: plans, the time-step driver and the
// host-buffer operators that stand in for the reference's gpu_*() functions.
//
// Reference behaviour followed (file:line under /root/reference/src/):
//   driver: buf0 <- padded input, buf1 <- 0, `times` launches ping-ponging, result = buf[times % 2],
//           timing = steady_clock around the launch loop + one device sync
//           (1d/gpu_1r.cu:103-134, 2d/gpu.cu:392-421, :450-479, :525-554, 3d/gpu_star.cu:158-192,
//            3d/gpu_box.cu:190-223)
//   stdout: label / "Time = <ms>[ms]" / "GStencil/s = %f" (e.g. 2d/gpu.cu:549-553)
// There is deliberately no CPU fallback: without a HIP device every compute entry point fails.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "engine.h"
#include "spans.h"

namespace lora {

static thread_local std::string g_last_error;
static thread_local int g_default_boundary = LORA_BC_REFERENCE;
static thread_local int g_default_normalize = 0;
static thread_local lora_run_info g_last_info = {};

void set_last_error(const char *what, hipError_t e) {
    g_last_error = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
}

void set_last_error_text(const char *text) { g_last_error = text ? text : ""; }
void set_last_run_info(const lora_run_info &info) { g_last_info = info; }

#define LORA_HIP_TRY(expr)                      \
    do {                                        \
        hipError_t e__ = (expr);                \
        if (e__ != hipSuccess) {                \
            lora::set_last_error(#expr, e__);   \
            return LORA_EHIP;                   \
        }                                       \
    } while (0)

static int outer_extent(const Plan &p) { return p.dims[0]; }

int region_granularity(const Plan &p) {
    switch (p.ndim) {
        case 1:
            return 2;
        default:
            return 1;  // 2D tiles and 3D chunks may start on any row / plane
    }
}

// Factors for the MFMA formulation out = sum_t (U_t X) V_t + residual taps, derived from the applied taps W:
//   * c x the star2d1r table  -> the reference's hard-coded rank-1 factor u = v = (0,1,2,4,2,1,0) plus its 8-point
//                                correction (2d/gpu.cu:486-487, :249-264), scaled by c;
//   * star-shaped W           -> vertical band = centre column, horizontal band = centre row without the centre
//                                (2d/gpu.cu:433-444), i.e. two terms with a unit factor each;
//   * anything else           -> pyramid factorisation into three terms (2d/gpu.cu:280-350); what it does not
//                                capture is applied as residual taps if it is sparse enough.
static void derive_lowrank(Plan &p) {
    LowRank2D &lr = p.lowrank;
    lr = LowRank2D{};
    p.lowrank_valid = false;
    if (p.ndim != 2) return;
    const double *W = p.w;
    auto finish_residual = [&](double tol) {
        double wmax = 0.0;
        for (int k = 0; k < 49; ++k) wmax = std::fmax(wmax, std::fabs(W[k]));
        lr.nresid = 0;
        for (int r = 0; r < 7; ++r)
            for (int c = 0; c < 7; ++c) {
                double s = 0.0;
                for (int t = 0; t < lr.rank; ++t) s += lr.u[t][r] * lr.v[t][c];
                const double d = W[r * 7 + c] - s;
                if (!std::isfinite(d)) return false;
                if (std::fabs(d) > tol * wmax) {
                    if (lr.nresid == 16) return false;
                    lr.rdy[lr.nresid] = r - 3;
                    lr.rdx[lr.nresid] = c - 3;
                    lr.rw[lr.nresid] = d;
                    ++lr.nresid;
                }
            }
        return true;
    };
    // (1) scaled star2d1r table
    double ref[49];
    default_params(LORA_STAR2D1R, ref);
    const double c = W[24] / ref[24];
    bool scaled = std::isfinite(c) && c != 0.0;
    for (int k = 0; k < 49 && scaled; ++k) scaled = std::fabs(W[k] - c * ref[k]) <= 1e-14 * std::fabs(c * ref[24]);
    if (scaled) {
        static const double f[7] = {0, 1, 2, 4, 2, 1, 0};
        lr.rank = 1;
        for (int e = 0; e < 7; ++e) {
            lr.u[0][e] = c * f[e];
            lr.v[0][e] = f[e];
        }
        p.lowrank_valid = finish_residual(1e-14);
        return;
    }
    // (2) star-shaped taps
    if (p.tapset == TAPS2D_STAR) {
        lr.rank = 2;
        for (int e = 0; e < 7; ++e) {
            lr.u[0][e] = W[e * 7 + 3];
            lr.v[0][e] = (e == 3) ? 1.0 : 0.0;
            lr.u[1][e] = (e == 3) ? 1.0 : 0.0;
            lr.v[1][e] = (e == 3) ? 0.0 : W[3 * 7 + e];
        }
        p.lowrank_valid = finish_residual(1e-14);
        return;
    }
    // (3) pyramid factorisation (the reference's scheme: exact for its symmetric tables)
    {
        double u[4][7], v[4][7];
        factorize_7x7(W, u, v, nullptr);
        lr.rank = 3;
        bool finite = true;
        for (int t = 0; t < 3; ++t)
            for (int e = 0; e < 7; ++e) {
                finite = finite && std::isfinite(u[t][e]) && std::isfinite(v[t][e]);
                lr.u[t][e] = u[t][e];
                lr.v[t][e] = v[t][e];
            }
        if (finite && finish_residual(1e-13)) {
            p.lowrank_valid = true;
            return;
        }
    }
    // (4) any other taps: truncated SVD, as many terms (<= 3) as the singular values ask for; what three terms do not
    //     capture must be a few isolated taps (applied on the vector pipe) or the taps are refused for this variant
    {
        double u[7][7], v[7][7], sigma[7];
        lr = LowRank2D{};
        if (svd_7x7(W, u, v, sigma) != LORA_OK || !(sigma[0] > 0.0)) return;
        int rank = 1;
        while (rank < 3 && sigma[rank] > 1e-14 * sigma[0]) ++rank;
        lr.rank = rank;
        for (int t = 0; t < rank; ++t)
            for (int e = 0; e < 7; ++e) {
                lr.u[t][e] = u[t][e];
                lr.v[t][e] = v[t][e];
            }
        p.lowrank_valid = finish_residual(1e-13);
    }
}

void plan_refresh(Plan &p) {
    ++p.epoch;
    if (p.ndim == 2) {
        // smallest tap set that covers the non-zero pattern of the applied taps
        bool diamond = true, star = true;
        for (int r = 0; r < 7; ++r)
            for (int c = 0; c < 7; ++c) {
                if (p.w[r * 7 + c] == 0.0) continue;
                const int ar = r < 3 ? 3 - r : r - 3, ac = c < 3 ? 3 - c : c - 3;
                if (ar != 0 && ac != 0) star = false;
                if (ar + ac > 3) diamond = false;
            }
        p.tapset = star ? TAPS2D_STAR : (diamond ? TAPS2D_DIAMOND : TAPS2D_BOX);
        derive_lowrank(p);
        if (p.variant == LORA_VARIANT_MFMA && !p.lowrank_valid) p.variant = LORA_VARIANT_DIRECT;
        // temporal fusion (two applications per launch) wins for every tap set once the tile height is tuned
        // (16384^2 / 8192^2, profiles/r01_sweep_fused_rows.jsonl: 25 taps 602 vs 352 GStencils/s, 13 taps 649 vs
        // 345, 49 taps 400 vs 350); the light 13-tap star prefers the small tile (more workgroups per CU), the
        // FMA-heavier sets the tall one (less recomputed halo)
        if (p.generic) {
            p.variant = LORA_VARIANT_DIRECT;
            p.lowrank_valid = false;
        }
        // odd innermost extent (rows only 8-byte aligned): the tiled kernels do not apply, but the row-streaming kernel
        // does -- its 16-byte row pieces need dword alignment only, and the last, half-valid column pair of a row is
        // cut by the store descriptor's per-dword range check -- so fused launches keep their speed and only the
        // single-sweep tail (at most one launch per run) goes through the generic kernel
        const bool odd_stream = p.generic && p.stream2 && p.boundary == LORA_BC_REFERENCE;
        if (p.variant == LORA_VARIANT_MFMA || (p.generic && !odd_stream))
            p.steps_per_launch = 1;
        else
            p.steps_per_launch = p.steps_per_launch_req == 0 ? 6 : p.steps_per_launch_req;
        // Temporal fusion wins for every tap set: star2d1r 16384^2 352 (one sweep per launch) -> 593 (tile kernel, 2)
        // -> 591 (row-streaming, 2) -> 843 GStencils/s (row-streaming, 4: profiles/r02_*).  Four applications per launch exist in the row-streaming kernel, reference boundary (the level-2 halo is the
        // source buffer's own; the Dirichlet option would need source rows 11 steps back)
        if (p.steps_per_launch == 4 && !(p.stream2 && p.boundary == LORA_BC_REFERENCE)) p.steps_per_launch = 2;
        // six: the workgroup-row kernel (kernels_2d_wg.hip), same conditions; otherwise four, otherwise two
        if (p.steps_per_launch == 6 && !(p.stream2 && p.boundary == LORA_BC_REFERENCE)) p.steps_per_launch = 2;
        p.fused_rows = p.fused_rows_req ? p.fused_rows_req : (p.tapset == TAPS2D_STAR ? 6 : 10);
        // Low-rank evaluation on the vector pipe inside the fused kernel (kernels_2d_fused.hip, apply_row): taken
        // when the factors have the support pattern one of its two forms is specialised for.
        p.fused_eval = p.tapset;
        p.lowrank_rc = 0.0;
        if (p.lowrank_valid && p.lowrank_valu != 0 && p.steps_per_launch >= 2) {
            const LowRank2D &lr = p.lowrank;
            auto outside_zero = [&](int t, int lo) {
                for (int e = 0; e < 7; ++e)
                    if ((e < lo || e > 6 - lo) && (lr.u[t][e] != 0.0 || lr.v[t][e] != 0.0)) return false;
                return true;
            };
            if (lr.rank == 1 && lr.nresid == 8 && outside_zero(0, 1)) {
                // residual must be +c on (+-3, 0), (0, +-3) and -c on (+-2, +-2)
                double c = 0.0;
                bool ok = true;
                for (int k = 0; k < 8 && ok; ++k) {
                    const int ay = lr.rdy[k] < 0 ? -lr.rdy[k] : lr.rdy[k], ax = lr.rdx[k] < 0 ? -lr.rdx[k] : lr.rdx[k];
                    const bool tip = (ay == 3 && ax == 0) || (ay == 0 && ax == 3), corner = ay == 2 && ax == 2;
                    if (!tip && !corner) ok = false;
                    const double ck = tip ? lr.rw[k] : -lr.rw[k];
                    if (k == 0) c = ck;
                    if (ck != c) ok = false;
                }
                if (ok) {
                    p.fused_eval = 3;
                    p.lowrank_rc = c;
                }
            } else if (lr.rank == 3 && lr.nresid == 0 && outside_zero(1, 1) && outside_zero(2, 2)) {
                p.fused_eval = 4;
                // mirror-symmetric horizontal profiles (every table the pyramid scheme accepts: it needs a symmetric
                // matrix): the mirrored taps of a row are pre-added once and shared by the three terms
                bool sym = p.lowrank_valu != 2;  // lowrank_valu = 2 keeps the plain pyramid form (A/B timing)
                for (int t = 0; t < 3 && sym; ++t)
                    for (int e = 0; e < 3; ++e)
                        if (lr.v[t][e] != lr.v[t][6 - e]) sym = false;
                if (sym) {
                    p.fused_eval = 5;
                    // the reference table's middle factor (0, 1, 0, -1, 0, 1, 0): zero taps skipped at compile time
                    if (p.lowrank_valu != 3 && lr.u[1][2] == 0.0 && lr.u[1][4] == 0.0 && lr.v[1][2] == 0.0) p.fused_eval = 6;
                }
            }
        }
        // Nested-profile form (rows_2d.h, EVAL_NEST): row dy of the table = g[k] T_k, k = 3 - |dy - 3|, T_0 = x3,
        // T_k = a_k T_(k-1) + (x_(3-k) + x_(3+k)).  Accepted only if the taps it implies are the plan's taps EXACTLY.
        if (p.lowrank_valu != 0 && p.lowrank_valu != 4 && p.steps_per_launch >= 2) {
            const double *W = p.w;
            double g[4], a[4] = {0, 0, 0, 0};
            bool ok = true;
            for (int k = 0; k < 4 && ok; ++k) {
                g[k] = W[k * 7 + (3 - k)];  // outermost tap of row k
                if (!(g[k] != 0.0) || !std::isfinite(g[k])) ok = false;
            }
            for (int k = 1; k < 4 && ok; ++k) {
                a[k] = W[k * 7 + (4 - k)] / g[k];  // next tap inwards / outermost tap
                if (!std::isfinite(a[k])) ok = false;
            }
            for (int dy = 0; dy < 7 && ok; ++dy) {
                const int k = dy <= 3 ? dy : 6 - dy;
                // coefficient of x_(3 +- e) in T_k: prod of a_j for j = e+1 .. k (1 for e = k, absent for e > k)
                for (int dx = 0; dx < 7; ++dx) {
                    const int e = dx <= 3 ? 3 - dx : dx - 3;
                    double c = 0.0;
                    if (e <= k) {
                        c = g[k];
                        for (int j2 = k; j2 > e; --j2) c *= a[j2];
                    }
                    if (c != W[dy * 7 + dx]) ok = false;
                }
            }
            if (ok) {
                p.fused_eval = 7;
                for (int k = 0; k < 4; ++k) {
                    p.nest_g[k] = g[k];
                    p.nest_a[k] = a[k];
                }
            }
        }
        // (49 direct taps -- a table with no low-rank form -- do not fit the scalar registers of the six-application kernel
        // beside its three levels per wave, and rounds 2 - 3 kept such plans at four applications per launch for it.
        // Measured, six win all the same: 395 against 264 - 295 GStencils/s at 8192^2, 432 against 333 at 16384^2
        // (tools/k6_general.py): the taps come from the constant cache, the bytes per sweep are a third less.)
        // which kernel family lora_plan_stepk launches: the workgroup-row kernel for six applications and, in such plans,
        // for the four- and two-application tails of a run; plans that ask for four or two keep the row-streaming kernel
        // (option wg = 1: the workgroup-row kernel at every depth)
        p.wg_active = p.stream2 && p.boundary == LORA_BC_REFERENCE && p.variant == LORA_VARIANT_DIRECT &&
                      (p.steps_per_launch == 6 || (p.wg == 1 && p.steps_per_launch >= 2)) && p.wg != 0;
        if (p.steps_per_launch == 6 && !p.wg_active) p.steps_per_launch = 4;
        // The Dirichlet option: the workgroup-row kernel at FOUR applications per launch (a row of halo values per level;
        // six would need 98 KB of LDS per workgroup), unless the plan asks for two or switches the kernel off
        if (p.boundary == LORA_BC_DIRICHLET && !p.generic && p.stream2 && p.wg != 0 && p.variant == LORA_VARIANT_DIRECT &&
            p.fused_eval != TAPS2D_BOX && (p.steps_per_launch_req == 0 || p.steps_per_launch_req >= 4)) {
            p.steps_per_launch = 4;
            p.wg_active = 1;
        }
        if (p.wg_active) prepare_2d_wg(p);  // kernel resolution + residency query now, not in the first launch of a run
        p.kernel_name = (p.generic && p.steps_per_launch == 1) ? kernel_name_generic(p)
                        : (p.variant == LORA_VARIANT_MFMA)
                            ? kernel_name_2d_mfma(p)
                            : (p.wg_active ? kernel_name_2d_wg(p)
                               : p.steps_per_launch >= 2 ? (p.stream2 ? kernel_name_2d_stream(p) : kernel_name_2d_fused2(p))
                                                       : kernel_name_2d_direct(p));
    } else if (p.ndim == 3) {
        bool star = true;
        for (int k = 0; k < 27; ++k) {
            if (p.w[k] == 0.0) continue;
            const int dz = k / 9, dy = (k / 3) % 3, dx = k % 3;
            if ((dz != 1) + (dy != 1) + (dx != 1) > 1) star = false;
        }
        p.tapset = star ? TAPS3D_STAR : TAPS3D_BOX;
        p.mfma3_valid = false;
        if (!star && p.dtype == LORA_BF16) {
            float w32[27], cba[9];
            for (int k = 0; k < 27; ++k) w32[k] = (float) p.w[k];
            if (separable_27(w32, cba)) {
                if (p.separable != 0) {
                    p.tapset = TAPS3D_SEP;
                    for (int k = 0; k < 9; ++k) p.sep[k] = cba[k];
                }
                p.mfma3_valid = mfma_factors_27(cba, &p.mfma3_scale, p.mfma3_abc) != 0;
            }
        }
        // the matrix-pipe variant exists for bf16 box taps with bf16-exact factors, reference boundary, fused launches
        if (p.variant == LORA_VARIANT_MFMA &&
            !(p.dtype == LORA_BF16 && p.mfma3_valid && p.boundary == LORA_BC_REFERENCE && p.steps_per_launch_req != 1))
            p.variant = LORA_VARIANT_DIRECT;
        // two applications per launch (kernels_3d_fused.hip): fp64 tiled path; default, as in 2D (star3d1r 512^3
        // 499 vs 288 GStencils/s, box3d1r 768^3 523 vs 300)
        p.steps_per_launch = (!p.generic && p.steps_per_launch_req != 1) ? 2 : 1;
        // fp64: THREE applications per launch in the plane-streaming kernel (kernels_3d_planes.hip) -- the grid is read
        // and written once per three sweeps.  An odd count needs no halo copies: launch k starts at global step 3 k and
        // runs on the reference's own buffer state (lora_plan_run)
        // Which fused kernel: the plane-streaming kernel needs a grid that fills its 60 x 60 tiles and 32-plane chunks a few
        // times over (tools/small3d.sh, GStencils/s per launch, tile kernel / planes K = 2 / planes K = 3: star 256^3 496 /
        // 452 / 371, 320^3 458 / 499 / 408, 448^3 562 / 627 / 545, 512^3 497 / 583 / 618, 768^3 530 / 603 / 722; box 256^3
        // 300 / 307, 320^3 307 / 351, 768^3 446 / 499): three applications from ~1.2e8 points (star), two from ~2.4e7,
        // the round-1 tile kernel below.  Option stream3: -1 this rule, 0 tile kernel, 1 plane-streaming kernel always;
        // steps_per_launch = 3 asks for it by itself.  (In its 27-tap order the box stays at two: the third level makes
        // the launch VALU- and LDS-bound; its separable form, below, takes three.)
        const bool planes_ok = p.dtype != LORA_BF16 && !p.generic && p.stream3 != 0 && p.boundary != LORA_BC_PERIODIC;
        const double npts = (double) p.dims[0] * p.dims[1] * p.dims[2];
        // Exactly separable fp64 box taps (the reference's: they depend on dx only) are evaluated as x / y / z passes in
        // the plane-streaming kernel, 9-10 instead of 27 multiply-adds per point (option separable = 0: the 27-tap order);
        // that makes a third application per launch pay for the box as well
        double cba64[9];
        const bool sep_ok = planes_ok && p.tapset == TAPS3D_BOX && p.separable != 0 && separable_27d(p.w, cba64) != 0;
        bool stream3 = false;
        if (planes_ok && p.steps_per_launch == 2) {
            const bool three_pays = p.tapset == TAPS3D_STAR || sep_ok;
            if (p.steps_per_launch_req == 3) {
                stream3 = true;
                p.steps_per_launch = 3;
            } else if (p.stream3 == 1 || npts >= (sep_ok ? 2.0e6 : 2.4e7)) {  // (separable box: from ~128^3, tools/rule3d.sh)
                stream3 = true;
                // (the separable box: 512^3 two applications 593, three 535-589; 768^3 588 / 673 GStencils/s)
                const double from = p.tapset == TAPS3D_STAR ? 1.2e8 : 3.0e8;
                if (p.steps_per_launch_req == 0 && three_pays && (p.stream3 == 1 || npts >= from)) p.steps_per_launch = 3;
            }
        }
        p.stream3_active = stream3 ? 1 : 0;
        p.sep64_valid = (stream3 && sep_ok) ? 1 : 0;
        // FOUR applications per launch with the levels in registers (kernels_3d_lanes.hip): fp64, the 7-point star or
        // exactly separable box taps, reference boundary, any extents (odd innermost ones too: its fused launches replace
        // the one-thread-per-point fallback, which then only serves single-sweep tails).  Big grids: a tile is 24 x 120
        // output points and a z-chunk re-reads 8 planes, so the launch wants ~256 tiles x long chunks (star3d1r
        // GStencils/s per launch, planes kernel / this one: see DESIGN 3.3d); option lanes3 = 1 / 0 forces either,
        // steps_per_launch = 4 asks for it by itself.
        double cba64l[9];
        const bool lanes_taps = p.dtype != LORA_BF16 && p.boundary == LORA_BC_REFERENCE &&
                                (p.tapset == TAPS3D_STAR ||
                                 (p.tapset == TAPS3D_BOX && p.separable != 0 && separable_27d(p.w, cba64l) != 0));
        bool lanes = false;
        if (lanes_taps && p.lanes3 != 0 && p.steps_per_launch_req != 1 && p.steps_per_launch_req != 2 &&
            p.steps_per_launch_req != 3)
            // (GStencils/s per launch, tile kernels (two per launch) against this one, tools/cube3d_check.py with the kernel's
            // second form: star 192^3 491 / 347, 224^3 498 / 552, 256^3 554 / 678, 320^3 507 / 881, 384^3 610 / 944; box
            // 224^3 504 / 441, 256^3 494 / 558, 320^3 512 / 765, 384^3 607 / 846; 256 x 512 x 128 444 / 532 and 414 / 442; odd
            // innermost extent 512 x 512 x 511: 126 (one thread per point) / 892.  Below ~10 M points there are too few
            // tiles x chunks for the one workgroup per CU this kernel runs.)
            lanes = p.steps_per_launch_req == 4 || p.lanes3 == 1 ||
                    npts >= (p.generic ? 1.0e7 : (p.tapset == TAPS3D_STAR ? 1.0e7 : 1.4e7));
        p.lanes3_active = lanes ? 1 : 0;
        if (lanes) {
            const int spl_before = p.steps_per_launch, stream3_before = p.stream3_active, sep_before = p.sep64_valid;
            p.steps_per_launch = 4;
            p.stream3_active = 0;
            if (p.tapset == TAPS3D_BOX) {
                p.sep64_valid = 1;
                for (int k = 0; k < 9; ++k) p.sep64[k] = cba64l[k];
            }
            if (!prepare_3d_lanes(p)) {  // the device has no room for a workgroup of it: the tile kernels stay
                lanes = false;
                p.lanes3_active = 0;
                p.steps_per_launch = spl_before;
                p.stream3_active = stream3_before;
                p.sep64_valid = sep_before;
            }
        }
        if (p.sep64_valid && !lanes)
            for (int k = 0; k < 9; ++k) p.sep64[k] = cba64[k];
        // bf16: FOUR applications per launch with the levels in registers (kernels_3d_bf16_lanes.hip): exactly separable box
        // taps on the vector pipe, reference boundary.  A tile is 56 x 120 output points on one 1024-thread workgroup per
        // CU, so the launch wants ~256 tiles x long chunks: by grid size (lanes3 = -1), never (0), always (1);
        // steps_per_launch = 4 asks for it by itself.
        bool blanes = false;
        if (p.dtype == LORA_BF16 && p.tapset == TAPS3D_SEP && p.boundary == LORA_BC_REFERENCE &&
            p.variant != LORA_VARIANT_MFMA && p.lanes3 != 0 && !p.generic &&
            (p.steps_per_launch_req == 0 || p.steps_per_launch_req == 4))
            // (GStencils/s per launch, this kernel / the two-sweep tile kernel, tools/bf16_crossover.py,
            // profiles/r04_bf16_lanes_crossover.jsonl: 192^3 395 / 501, 256^3 891 / 788, 320^3 1487 / 1104, 512^3 1795 / 1345,
            // 48 x 768^2 1243 / 1123, 96 x 768^2 1561 / 1397, 768^3 2090 / 1662)
            blanes = (p.steps_per_launch_req == 4 || p.lanes3 == 1 || npts >= 1.2e7) && prepare_3d_bf16_lanes(p);
        if (blanes) {
            p.lanes3_active = 1;
            p.steps_per_launch = 4;
        }
        p.kernel_name = blanes ? kernel_name_3d_bf16_lanes(p)
                        : (p.dtype == LORA_BF16)
                            ? (p.steps_per_launch == 2 ? (p.variant == LORA_VARIANT_MFMA ? kernel_name_3d_bf16_mfma2(p)
                                                                                          : kernel_name_3d_bf16_fused2(p))
                                                       : kernel_name_3d_bf16(p))
                        : lanes                ? kernel_name_3d_lanes(p)
                        : p.generic            ? kernel_name_generic(p)
                        : p.steps_per_launch >= 2 ? (stream3 ? kernel_name_3d_stream(p) : kernel_name_3d_fused2(p))
                                                  : kernel_name_3d(p);
    } else {
        p.tapset = 0;
        // K applications per launch (kernels_1d.hip): 8 by default, the single sweep being launch-latency bound
        p.steps_per_launch = p.steps_per_launch_req == 0 ? 8 : p.steps_per_launch_req;
        p.kernel_name = p.steps_per_launch > 1 ? kernel_name_1d_fused(p) : kernel_name_1d(p);
    }
}

int check_buffers(const void *a, const void *b) {
    if (!a || !b) return LORA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(b) & 15)) {
        g_last_error = "device buffers must be 16-byte aligned";
        return LORA_EUNSUPPORTED;
    }
    return LORA_OK;
}

static int step_region(Plan &p, const void *d_in, void *d_out, int begin, int end, hipStream_t s) {
    if (int rc = check_buffers(d_in, d_out)) return rc;
    if (d_in == d_out) return LORA_EINVAL;
    const int ext = outer_extent(p);
    if (begin < 0 || end > ext || begin > end) return LORA_EINVAL;
    const int g = region_granularity(p);
    if (begin % g != 0) return LORA_EINVAL;
    const double *in = static_cast<const double *>(d_in);
    double *out = static_cast<double *>(d_out);
    hipError_t e;
    if (p.dtype == LORA_BF16)
        e = launch_3d_bf16(p, d_in, d_out, begin, end, s);
    else if (p.generic)
        e = (p.ndim == 2) ? launch_2d_generic(p, in, out, begin, end, s) : launch_3d_generic(p, in, out, begin, end, s);
    else if (p.ndim == 1)
        e = launch_1d(p, in, out, begin, end, s);
    else if (p.ndim == 2)
        e = (p.variant == LORA_VARIANT_MFMA) ? launch_2d_mfma(p, in, out, begin, end, s)
                                             : launch_2d_direct(p, in, out, begin, end, s);
    else
        e = launch_3d(p, in, out, begin, end, s);
    if (e != hipSuccess) {
        set_last_error("kernel launch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

// Three applications (3D fp64, plane-streaming kernel) starting at a global step of the given parity; `d_halo` is a
// buffer that carries the caller's halo (what even global levels see outside the interior).
static int step3_natural(Plan &p, const void *d_in, void *d_out, const void *d_halo, int parity, int begin, int end,
                         void *stream) {
    if (!(p.ndim == 3 && p.dtype != LORA_BF16 && !p.generic && p.stream3_active && p.steps_per_launch == 3))
        return LORA_EUNSUPPORTED;
    if (int rc = check_buffers(d_in, d_out)) return rc;
    if (d_in == d_out || !d_halo || begin < 0 || end > p.dims[0] || begin > end) return LORA_EINVAL;
    const hipError_t e = launch_3d_stream(p, 3, static_cast<const double *>(d_in), static_cast<double *>(d_out),
                                          static_cast<const double *>(d_halo), parity, begin, end,
                                          static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_last_error("fused 3-step kernel launch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

}  // namespace lora

using lora::g_default_boundary;
using lora::g_default_normalize;
using lora::g_last_error;
using lora::Plan;

extern "C" {

const char *lora_strerror(int status) {
    switch (status) {
        case LORA_OK:
            return "ok";
        case LORA_EINVAL:
            return "invalid argument";
        case LORA_EUNSUPPORTED:
            return "unsupported size or alignment";
        case LORA_EHIP:
            return "HIP runtime error";
        case LORA_ENOMEM:
            return "out of memory";
        case LORA_ENODEVICE:
            return "no HIP device (this engine has no CPU fallback)";
        default:
            return "unknown status";
    }
}

const char *lora_last_error(void) { return g_last_error.c_str(); }

int lora_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void) hipGetLastError();
        return 0;
    }
    return n;
}

int lora_plan_create(lora_plan **out, int shape, int dtype, const int *dims, const double *params) {
    if (!out || !dims) return LORA_EINVAL;
    *out = nullptr;
    const int nd = lora::shape_ndim(shape);
    if (nd == 0 || (dtype != LORA_F64 && dtype != LORA_BF16)) return LORA_EINVAL;
    if (dtype == LORA_BF16 && nd != 3) {
        g_last_error = "bf16 is implemented for the 3D shapes only";
        return LORA_EUNSUPPORTED;
    }
    if (dtype == LORA_BF16 && (dims[2] & 7)) {
        g_last_error = "bf16 grids need an innermost extent that is a multiple of 8";
        return LORA_EUNSUPPORTED;
    }
    for (int d = 0; d < nd; ++d)
        if (dims[d] <= 0) return LORA_EINVAL;
    // 2D/3D rows are read and written in 16-byte pieces by the tiled kernels; an odd innermost extent falls back to
    // the generic one-thread-per-point kernels (fp64 only)
    const bool odd_inner = nd >= 2 && (dims[nd - 1] & 1);
    if (odd_inner && dtype != LORA_F64) {
        g_last_error = "innermost extent must be even";
        return LORA_EUNSUPPORTED;
    }
    if ((double) lora_padded_count(shape, dims) >= 2147483647.0 * 64) return LORA_EUNSUPPORTED;
    if (nd == 1 && dims[0] > 2147483647 - 8) {
        g_last_error = "1D extent too large (kernels index the padded array with 32-bit integers)";
        return LORA_EUNSUPPORTED;
    }
    lora_plan *pl = new (std::nothrow) lora_plan();
    if (!pl) return LORA_ENOMEM;
    Plan &p = pl->p;
    p.shape = shape;
    p.ndim = nd;
    p.dtype = dtype;
    for (int d = 0; d < nd; ++d) p.dims[d] = dims[d];
    p.ntaps = lora::shape_ntaps(shape);
    double tmp[49];
    if (!params) {
        lora::default_params(shape, tmp);
        params = tmp;
    }
    lora::effective_weights(shape, params, p.w);
    if (g_default_normalize) {  // normalised-weights mode: the operator's taps divided by their sum
        double sum = 0.0;
        for (int k = 0; k < p.ntaps; ++k) sum += p.w[k];
        if (sum != 0.0 && std::isfinite(sum))
            for (int k = 0; k < p.ntaps; ++k) p.w[k] /= sum;
    }
    p.variant = LORA_VARIANT_DIRECT;
    p.generic = odd_inner;
    p.boundary = g_default_boundary;
    if (nd == 3) {
        // enough workgroups to fill 256 CUs a few times over, chunks as long as that allows
        const long tiles = (long) ((dims[2] + 127) / 128) * ((dims[1] + 15) / 16);
        int zc = 16;
        while (zc > 4 && tiles * ((dims[0] + zc - 1) / zc) < 2048) zc = (zc == 16) ? 7 : 4;
        p.z_chunk = zc;
    }
    lora::plan_refresh(p);
    if (p.boundary == LORA_BC_PERIODIC && lora_plan_set_boundary(pl, LORA_BC_PERIODIC) != LORA_OK) {
        delete pl;
        return LORA_EUNSUPPORTED;
    }
    *out = pl;
    return LORA_OK;
}

static void torus_drop(lora_plan *plan);

void lora_plan_destroy(lora_plan *plan) {
    if (plan && plan->graph_exec) (void) hipGraphExecDestroy(plan->graph_exec);
    if (plan && plan->scratch) (void) hipFree(plan->scratch);
    if (plan) torus_drop(plan);
    delete plan;
}

int lora_plan_set_weights(lora_plan *plan, const double *weights, int count) {
    if (!plan || !weights || count != plan->p.ntaps) return LORA_EINVAL;
    std::memcpy(plan->p.w, weights, sizeof(double) * count);
    lora::plan_refresh(plan->p);
    return LORA_OK;
}

int lora_plan_get_weights(const lora_plan *plan, double *weights, int count) {
    if (!plan || !weights || count != plan->p.ntaps) return LORA_EINVAL;
    std::memcpy(weights, plan->p.w, sizeof(double) * count);
    return LORA_OK;
}

int lora_set_default_boundary(int boundary) {
    const int old = g_default_boundary;
    if (boundary >= LORA_BC_REFERENCE && boundary <= LORA_BC_PERIODIC) g_default_boundary = boundary;
    return old;
}

int lora_set_default_normalize(int on) {
    const int old = g_default_normalize;
    g_default_normalize = on ? 1 : 0;
    return old;
}

int lora_plan_set_boundary(lora_plan *plan, int boundary) {
    if (!plan || boundary < LORA_BC_REFERENCE || boundary > LORA_BC_PERIODIC) return LORA_EINVAL;
    if (boundary == LORA_BC_PERIODIC) {
        static const int h1[1] = {4}, h2[2] = {4, 4}, h3[3] = {1, 2, 4};
        const int *h = plan->p.ndim == 1 ? h1 : (plan->p.ndim == 2 ? h2 : h3);
        for (int d = 0; d < plan->p.ndim; ++d)
            if (plan->p.dims[d] < h[d]) {
                g_last_error = "periodic boundary needs every extent >= its halo width";
                return LORA_EUNSUPPORTED;
            }
    }
    plan->p.boundary = boundary;
    lora::plan_refresh(plan->p);  // the boundary option decides how many applications a launch may fuse
    return LORA_OK;
}

int lora_plan_set_variant(lora_plan *plan, int variant) {
    if (!plan) return LORA_EINVAL;
    if (variant == LORA_VARIANT_AUTO) variant = LORA_VARIANT_DIRECT;
    if (variant != LORA_VARIANT_DIRECT && variant != LORA_VARIANT_MFMA) return LORA_EINVAL;
    if (variant == LORA_VARIANT_MFMA && plan->p.ndim == 3) {
        // bf16 box taps: in-plane passes on v_mfma_f32_16x16x32_bf16 (kernels_3d_bf16_mfma.hip)
        const Plan &q = plan->p;
        if (!(q.dtype == LORA_BF16 && q.mfma3_valid && q.boundary == LORA_BC_REFERENCE && q.steps_per_launch_req != 1)) {
            g_last_error = "the bf16 MFMA variant takes separable box taps with bf16-exact factors, reference boundary, fused launches";
            return LORA_EUNSUPPORTED;
        }
    } else if (variant == LORA_VARIANT_MFMA && plan->p.ndim != 2) {
        return LORA_EUNSUPPORTED;
    } else if (variant == LORA_VARIANT_MFMA && !plan->p.lowrank_valid) {
        g_last_error = "these taps have no rank<=3 + sparse-residual factorisation";
        return LORA_EUNSUPPORTED;
    }
    plan->p.variant = variant;
    lora::plan_refresh(plan->p);
    return LORA_OK;
}

int lora_plan_set_option(lora_plan *plan, const char *key, int value) {
    if (!plan || !key) return LORA_EINVAL;
    Plan &p = plan->p;
    if (!std::strcmp(key, "rows_per_thread")) {
        if (value != 4 && value != 8 && value != 16) return LORA_EINVAL;
        p.rows_per_thread = value;
    } else if (!std::strcmp(key, "panel_width")) {
        if (value < 1) return LORA_EINVAL;
        p.panel_width = value;
    } else if (!std::strcmp(key, "z_chunk")) {
        if (value < 1) return LORA_EINVAL;
        p.z_chunk = value;
    } else if (!std::strcmp(key, "nt_store")) {
        p.nt_store = value ? 1 : 0;
    } else if (!std::strcmp(key, "persistent")) {
        p.persistent = value ? 1 : 0;
    } else if (!std::strcmp(key, "stream")) {
        p.stream2 = value ? 1 : 0;
    } else if (!std::strcmp(key, "stream_rows")) {
        if (value < 0 || value > (1 << 20)) return LORA_EINVAL;
        p.stream_rows = value;
    } else if (!std::strcmp(key, "wg")) {
        if (value < -1 || value > 1) return LORA_EINVAL;
        p.wg = value;
    } else if (!std::strcmp(key, "wg_rows")) {
        if (value < 0 || value > (1 << 20)) return LORA_EINVAL;
        p.wg_rows = value;
    } else if (!std::strcmp(key, "wg_prio")) {
        if (value < 0 || value > 24) return LORA_EINVAL;
        p.wg_prio = value;
    } else if (!std::strcmp(key, "wg_edge_pct")) {
        if (value < -1 || value > 100) return LORA_EINVAL;
        p.wg_edge_pct = value;
    } else if (!std::strcmp(key, "stream_depth")) {
        if (value < 2 || value > 6) return LORA_EINVAL;
        p.stream_depth = value;
    } else if (!std::strcmp(key, "stream3")) {
        if (value < -1 || value > 1) return LORA_EINVAL;
        p.stream3 = value;
    } else if (!std::strcmp(key, "lanes3")) {
        if (value < -1 || value > 1) return LORA_EINVAL;
        p.lanes3 = value;
    } else if (!std::strcmp(key, "stream3_waves")) {
        if (value != 0 && value != 4 && value != 8) return LORA_EINVAL;
        p.stream3_waves = value;
    } else if (!std::strcmp(key, "stream3_async")) {
        p.stream3_async = value ? 1 : 0;
    } else if (!std::strcmp(key, "stream3_pipe")) {
        p.stream3_pipe = value ? 1 : 0;
    } else if (!std::strcmp(key, "stream3_slots")) {
        if (value != 0 && value != 2) return LORA_EINVAL;  // the ring has two slots (deeper ones measured, no gain)
        p.stream3_slots = value;
    } else if (!std::strcmp(key, "stream_share")) {
        p.stream_share = value ? 1 : 0;
    } else if (!std::strcmp(key, "stream_prefetch")) {
        p.stream_prefetch = value ? 1 : 0;
    } else if (!std::strcmp(key, "stream_sync")) {
        if (value < 0 || value > 2) return LORA_EINVAL;
        p.stream_sync = value;
    } else if (!std::strcmp(key, "scratch")) {
        if (value < -1 || value > 1) return LORA_EINVAL;
        p.use_scratch = value;
    } else if (!std::strcmp(key, "mfma_split")) {
        p.mfma_split = value ? 1 : 0;
    } else if (!std::strcmp(key, "graph")) {
        if (value < -1 || value > 1) return LORA_EINVAL;
        p.use_graph = value;
    } else if (!std::strcmp(key, "lowrank_valu")) {
        if (value < -1 || value > 4) return LORA_EINVAL;  // 2 / 3: plain / symmetric pyramid form, 4: rank-1 + correction
                                                           // instead of the nested-profile form (A/B timing)
        p.lowrank_valu = value;
    } else if (!std::strcmp(key, "separable")) {
        if (value < -1 || value > 1) return LORA_EINVAL;
        p.separable = value;
    } else if (!std::strcmp(key, "ablate")) {
#ifdef LORA_DIAGNOSTICS
        p.ablate = value & 63;
#else
        g_last_error = "option \"ablate\" exists only in -DLORA_DIAGNOSTICS builds";
        return LORA_EINVAL;  // wrong-results timing experiments are not part of the shipped library
#endif
    } else if (!std::strcmp(key, "lds_dma")) {
        p.lds_dma = value ? 1 : 0;
    } else if (!std::strcmp(key, "cols_per_lane")) {
        if (value != 4 && value != 8) return LORA_EINVAL;
        p.cols_per_lane = value;
    } else if (!std::strcmp(key, "fused_rows")) {
        if (value != 0 && value != 6 && value != 8 && value != 10) return LORA_EINVAL;
        p.fused_rows_req = value;
    } else if (!std::strcmp(key, "steps_per_launch")) {
        const bool three = value == 3 && p.ndim == 3 && p.dtype != LORA_BF16;  // 3D fp64 plane-streaming kernel
        const bool six = value == 6 && p.ndim == 2;                             // 2D workgroup-row kernel
        const bool four3 = value == 4 && p.ndim == 3;  // 3D register-resident kernels (fp64: star / separable box; bf16: separable box)
        if (value < 0 || value > 32 || ((value & (value - 1)) && !three && !six)) return LORA_EINVAL;  // 0 (auto), 1, 2, 4, 8, 16, 32
        if (value > 8 && p.ndim != 1) return LORA_EUNSUPPORTED;  // 16 and 32 exist in 1D only
        const bool fusable = (p.ndim == 2 && p.variant == LORA_VARIANT_DIRECT && (!p.generic || p.stream2)) ||
                             (p.ndim == 3 && (!p.generic || four3)) || p.ndim == 1;
        if (value >= 2 && !fusable) return LORA_EUNSUPPORTED;
        if (value > 2 && p.ndim == 3 && !three && !four3) return LORA_EUNSUPPORTED;  // 3D: two; three (plane-streaming) or four (register-resident) in fp64
        if (value > 4 && p.ndim == 2 && !six) return LORA_EUNSUPPORTED;  // 2D: two, four (row-streaming kernel), six (workgroup rows)
        p.steps_per_launch_req = value;
    } else if (!std::strcmp(key, "fused_pipeline")) {
        p.fused_pipeline = value ? 1 : 0;
    } else if (!std::strcmp(key, "fused_z_chunk")) {
        if (value < 0 || value > 4096) return LORA_EINVAL;
        p.fused_z_chunk = value;
    } else if (!std::strcmp(key, "spans3")) {
        if (value < -1 || value > 2) return LORA_EINVAL;
        p.spans3 = value;
    } else if (!std::strcmp(key, "torus")) {
        p.torus = value ? 1 : 0;
    } else {
        return LORA_EINVAL;
    }
    lora::plan_refresh(p);
    return LORA_OK;
}

int lora_plan_get_option(const lora_plan *plan, const char *key, int *value) {
    if (!plan || !key || !value) return LORA_EINVAL;
    const Plan &p = plan->p;
    if (!std::strcmp(key, "rows_per_thread"))
        *value = p.rows_per_thread;
    else if (!std::strcmp(key, "panel_width"))
        *value = p.panel_width;
    else if (!std::strcmp(key, "z_chunk"))
        *value = p.z_chunk;
    else if (!std::strcmp(key, "nt_store"))
        *value = p.nt_store;
    else if (!std::strcmp(key, "persistent"))
        *value = p.persistent;
    else if (!std::strcmp(key, "graph"))
        *value = p.use_graph;
    else if (!std::strcmp(key, "stream"))
        *value = p.stream2;
    else if (!std::strcmp(key, "stream3"))
        *value = p.stream3;
    else if (!std::strcmp(key, "stream3_waves"))
        *value = p.stream3_waves;
    else if (!std::strcmp(key, "stream3_slots"))
        *value = p.stream3_slots;
    else if (!std::strcmp(key, "stream3_pipe"))
        *value = p.stream3_pipe;
    else if (!std::strcmp(key, "stream3_async"))
        *value = p.stream3_async;
    else if (!std::strcmp(key, "stream_rows"))
        *value = p.stream_rows;
    else if (!std::strcmp(key, "stream_depth"))
        *value = p.stream_depth;
    else if (!std::strcmp(key, "stream_sync"))
        *value = p.stream_sync;
    else if (!std::strcmp(key, "boundary"))
        *value = p.boundary;
    else if (!std::strcmp(key, "fused_eval"))
        *value = p.fused_eval;
    else if (!std::strcmp(key, "lowrank_valu"))
        *value = p.lowrank_valu;
    else if (!std::strcmp(key, "lds_dma"))
        *value = p.lds_dma;
    else if (!std::strcmp(key, "separable"))
        *value = p.separable;
    else if (!std::strcmp(key, "cols_per_lane"))
        *value = p.cols_per_lane;
    else if (!std::strcmp(key, "fused_rows"))
        *value = p.fused_rows;
    else if (!std::strcmp(key, "steps_per_launch"))
        *value = p.steps_per_launch;
    else if (!std::strcmp(key, "fused_z_chunk"))
        *value = p.fused_z_chunk;
    else if (!std::strcmp(key, "spans3"))
        *value = p.spans3;
    else if (!std::strcmp(key, "torus"))
        *value = p.torus;
    else if (!std::strcmp(key, "fused_pipeline"))
        *value = p.fused_pipeline;
    else if (!std::strcmp(key, "tapset"))
        *value = p.tapset;
    else if (!std::strcmp(key, "variant"))
        *value = p.variant;
    else
        return LORA_EINVAL;
    return LORA_OK;
}

size_t lora_plan_padded_bytes(const lora_plan *plan) {
    if (!plan) return 0;
    return lora_padded_count(plan->p.shape, plan->p.dims) * (plan->p.dtype == LORA_BF16 ? 2 : sizeof(double));
}

const char *lora_plan_kernel_name(const lora_plan *plan) { return plan ? plan->p.kernel_name.c_str() : ""; }

const char *lora_plan_kernel_signature(const lora_plan *plan) {
    if (!plan) return "";
    const Plan &p = plan->p;
    static thread_local std::string sig;
    char buf[256];
    const std::string &k = p.kernel_name;
    buf[0] = 0;
    if (k == "stencil2d_stream_kernel") {
        const int K = p.steps_per_launch, w = lora::stream_strip_width(K);
        const int depth = p.boundary == LORA_BC_DIRICHLET ? 4 : (K == 4 ? (p.stream_depth == 2 ? 2 : 3) : p.stream_depth);
        std::snprintf(buf, sizeof buf, "eval=%d,k=%d,depth=%d,sync=%d%s,rows=%d,bc=%d", p.fused_eval, K, depth,
                      p.stream_sync, p.stream_share ? ",share=1" : ((K == 4 && p.stream_sync == 1 && p.stream_prefetch) ? ",pf=1" : ""), lora::stream_rows_per_chunk(p, K, p.dims[0], (p.dims[1] + w - 1) / w), p.boundary);
    }
    else if (k == "stencil2d_wg_kernel")
        std::snprintf(buf, sizeof buf, "eval=%d,k=%d,rows=%d,edge=%d,prio=%d,bc=%d", p.fused_eval, p.steps_per_launch, p.wg_rows,
                      p.wg_edge_pct, p.wg_prio, p.boundary);
    else if (k == "stencil2d_fused2_kernel")
        std::snprintf(buf, sizeof buf, "eval=%d,rows=%d,persist=%d,panel=%d,bc=%d", p.fused_eval, p.fused_rows,
                      p.persistent, p.panel_width, p.boundary);
    else if (k == "stencil2d_direct_kernel")
        std::snprintf(buf, sizeof buf, "taps=%d,rpt=%d,nt=%d,panel=%d", p.tapset, p.rows_per_thread, p.nt_store,
                      p.panel_width);
    else if (k == "stencil2d_mfma_kernel")
        std::snprintf(buf, sizeof buf, "rank=%d,panel=%d", p.lowrank.rank, p.panel_width);
    else if (k == "stencil3d_lanes_kernel")
        std::snprintf(buf, sizeof buf, "taps=%d,k=%d,fzc=%d,sp=%d,bc=%d", p.sep64_valid ? 2 : p.tapset, p.steps_per_launch,
                      p.fused_z_chunk, p.spans3, p.boundary);
    else if (k == "stencil3d_bf16_lanes_kernel")
        std::snprintf(buf, sizeof buf, "taps=%d,k=%d,fzc=%d,sp=%d,bc=%d", p.tapset, p.steps_per_launch, p.fused_z_chunk, p.spans3,
                      p.boundary);
    else if (k == "stencil3d_planes_kernel")
{
        const int K = p.steps_per_launch, pipe = (K == 2 || p.stream3_pipe) ? 1 : 0;
        const int nw = lora::stream3_waves(p, K, pipe);
        if (p.stream3_async && (nw == 8 || nw == 4))  // the launcher's own condition (kernels_3d_planes.hip)
            std::snprintf(buf, sizeof buf, "taps=%d,k=%d,waves=%d,async=1,fzc=%d,bc=%d", p.tapset, K, nw, p.fused_z_chunk,
                          p.boundary);
        else
            std::snprintf(buf, sizeof buf, "taps=%d,k=%d,waves=%d,slots=%d,pipe=%d,fzc=%d,bc=%d", p.sep64_valid ? 2 : p.tapset, K,
                          nw, lora::stream3_slots(K, nw, pipe, p.stream3_slots), pipe, p.fused_z_chunk, p.boundary);
    }
    else if (p.ndim == 3 && p.dtype == LORA_BF16)
        std::snprintf(buf, sizeof buf, "taps=%d,zc=%d,fzc=%d,cpl=%d,dma=%d,pipe=%d,bc=%d", p.tapset, p.z_chunk,
                      p.fused_z_chunk, p.cols_per_lane, p.lds_dma, p.fused_pipeline, p.boundary);
    else if (p.ndim == 3)
        std::snprintf(buf, sizeof buf, "taps=%d,zc=%d,fzc=%d,bc=%d", p.tapset, p.z_chunk, p.fused_z_chunk, p.boundary);
    else if (p.ndim == 1)
        std::snprintf(buf, sizeof buf, "k=%d", p.steps_per_launch);
    sig = k + "[" + buf + "]";
    return sig.c_str();
}

int lora_plan_region_granularity(const lora_plan *plan) { return plan ? lora::region_granularity(plan->p) : 0; }

int lora_plan_step_region(lora_plan *plan, const void *d_in, void *d_out, int begin, int end, void *stream) {
    if (!plan) return LORA_EINVAL;
    return lora::step_region(plan->p, d_in, d_out, begin, end, static_cast<hipStream_t>(stream));
}

int lora_plan_step(lora_plan *plan, const void *d_in, void *d_out, void *stream) {
    if (!plan) return LORA_EINVAL;
    return lora::step_region(plan->p, d_in, d_out, 0, plan->p.dims[0], static_cast<hipStream_t>(stream));
}

int lora_plan_step2_region(lora_plan *plan, const void *d_in, void *d_out, int begin, int end, void *stream) {
    if (!plan) return LORA_EINVAL;
    Plan &p = plan->p;
    const bool ok2 = p.ndim == 2 && p.variant == LORA_VARIANT_DIRECT &&
                     (!p.generic || (p.stream2 && p.boundary == LORA_BC_REFERENCE));
    const bool ok3 = p.ndim == 3 && !p.generic;
    if (p.ndim == 3 && p.lanes3_active) {  // the two-application tail of a four-application plan (also: odd innermost extents)
        if (int rc = lora::check_buffers(d_in, d_out)) return rc;
        if (d_in == d_out || begin < 0 || end > p.dims[0] || begin > end) return LORA_EINVAL;
        // Two sweeps move the same bytes as four, and at that the plane-streaming kernel is the faster of the two (star3d1r
        // 512^3: 475 against 551 us; box3d1r 768^3: 1651 against 1893; tools/tail_time.py) -- same taps in the same order,
        // same bits.  Deep regions of even-extent fp64 grids take it; the rest (thin regions, odd extents, bf16) stays here.
        if (p.dtype != LORA_BF16 && !p.generic && p.stream3 != 0 && end - begin >= 128) {
            Plan q = p;
            q.lanes3_active = 0;
            q.stream3_active = 1;
            q.steps_per_launch = 2;
            const hipError_t e = lora::launch_3d_stream(q, 2, static_cast<const double *>(d_in), static_cast<double *>(d_out),
                                                        static_cast<const double *>(d_in), 0, begin, end, static_cast<hipStream_t>(stream));
            if (e != hipSuccess) {
                lora::set_last_error("fused plane-streaming kernel launch (two-application tail)", e);
                return LORA_EHIP;
            }
            return LORA_OK;
        }
        const hipError_t e = p.dtype == LORA_BF16
                                 ? lora::launch_3d_bf16_lanes(p, 2, d_in, d_out, begin, end, static_cast<hipStream_t>(stream))
                                 : lora::launch_3d_lanes(p, 2, static_cast<const double *>(d_in), static_cast<double *>(d_out), begin,
                                                         end, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) {
            lora::set_last_error("fused register-resident 3D kernel launch", e);
            return LORA_EHIP;
        }
        return LORA_OK;
    }
    if (!ok2 && !ok3) return LORA_EUNSUPPORTED;
    if (int rc = lora::check_buffers(d_in, d_out)) return rc;
    if (d_in == d_out || begin < 0 || end > p.dims[0] || begin > end) return LORA_EINVAL;
    const hipError_t e = ok2 ? (p.stream2 ? lora::launch_2d_stream(p, 2, static_cast<const double *>(d_in),
                                                                   static_cast<double *>(d_out), begin, end,
                                                                   static_cast<hipStream_t>(stream))
                                          : lora::launch_2d_fused2(p, static_cast<const double *>(d_in),
                                                                   static_cast<double *>(d_out), begin, end,
                                                                   static_cast<hipStream_t>(stream)))
                         : p.dtype == LORA_BF16
                             ? (p.variant == LORA_VARIANT_MFMA
                                    ? lora::launch_3d_bf16_mfma2(p, d_in, d_out, begin, end, static_cast<hipStream_t>(stream))
                                    : lora::launch_3d_bf16_fused2(p, d_in, d_out, begin, end, static_cast<hipStream_t>(stream)))
                         : p.stream3_active
                             ? lora::launch_3d_stream(p, 2, static_cast<const double *>(d_in), static_cast<double *>(d_out),
                                                      static_cast<const double *>(d_in), 0, begin, end,
                                                      static_cast<hipStream_t>(stream))
                             : lora::launch_3d_fused2(p, static_cast<const double *>(d_in), static_cast<double *>(d_out),
                                                      begin, end, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        lora::set_last_error("fused kernel launch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

int lora_copy_block_f64(void *d_dst, long dst_ld, const void *d_src, long src_ld, long rows, long cols, void *stream) {
    if (!d_dst || !d_src || rows < 0 || cols < 0 || dst_ld < cols || src_ld < cols) return LORA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_src)) & 7) return LORA_EUNSUPPORTED;
    const hipError_t e = lora::launch_copy_block(static_cast<double *>(d_dst), dst_ld, static_cast<const double *>(d_src), src_ld,
                                                 rows, cols, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        lora::set_last_error("block copy kernel launch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

int lora_plan_halo(lora_plan *plan, void *d_dst, const void *d_src, int mode, void *stream) {
    if (!plan || !d_dst) return LORA_EINVAL;
    if (mode != LORA_HALO_COPY && mode != LORA_HALO_ZERO && mode != LORA_HALO_WRAP) return LORA_EINVAL;
    if (mode == LORA_HALO_COPY && (!d_src || d_src == d_dst)) return LORA_EINVAL;
    const Plan &p = plan->p;
    if (mode == LORA_HALO_WRAP) {
        static const int h1[1] = {4}, h2[2] = {4, 4}, h3[3] = {1, 2, 4};
        const int *h = p.ndim == 1 ? h1 : (p.ndim == 2 ? h2 : h3);
        for (int d = 0; d < p.ndim; ++d)
            if (p.dims[d] < h[d]) return LORA_EUNSUPPORTED;  // the wrap source would be a halo cell itself
    }
    const hipError_t e = lora::launch_halo(p, d_dst, d_src, mode, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        lora::set_last_error("halo kernel launch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

int lora_plan_step2(lora_plan *plan, const void *d_in, void *d_out, void *stream) {
    if (!plan) return LORA_EINVAL;
    return lora_plan_step2_region(plan, d_in, d_out, 0, plan->p.dims[0], stream);
}

int lora_plan_stepk_region(lora_plan *plan, const void *d_in, void *d_out, int begin, int end, void *stream) {
    if (!plan) return LORA_EINVAL;
    Plan &p = plan->p;
    if (p.steps_per_launch <= 1) return lora_plan_step_region(plan, d_in, d_out, begin, end, stream);
    if (p.ndim == 2 && (p.steps_per_launch == 6 || (p.wg_active && (p.steps_per_launch == 4 || p.steps_per_launch == 2)))) {
        if (int rc = lora::check_buffers(d_in, d_out)) return rc;
        if (d_in == d_out || begin < 0 || end > p.dims[0] || begin > end) return LORA_EINVAL;
        const hipError_t e = lora::launch_2d_wg(p, p.steps_per_launch, static_cast<const double *>(d_in),
                                                static_cast<double *>(d_out), begin, end, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) {
            lora::set_last_error("fused workgroup-row kernel launch", e);
            return LORA_EHIP;
        }
        return LORA_OK;
    }
    if (p.ndim == 2 && p.steps_per_launch == 4) {
        if (int rc = lora::check_buffers(d_in, d_out)) return rc;
        if (d_in == d_out || begin < 0 || end > p.dims[0] || begin > end) return LORA_EINVAL;
        const hipError_t e = lora::launch_2d_stream(p, 4, static_cast<const double *>(d_in), static_cast<double *>(d_out),
                                                    begin, end, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) {
            lora::set_last_error("fused 4-step kernel launch", e);
            return LORA_EHIP;
        }
        return LORA_OK;
    }
    if (p.ndim == 3 && p.lanes3_active && (p.steps_per_launch == 4 || p.steps_per_launch == 2)) {
        if (int rc = lora::check_buffers(d_in, d_out)) return rc;
        if (d_in == d_out || begin < 0 || end > p.dims[0] || begin > end) return LORA_EINVAL;
        const hipError_t e = p.dtype == LORA_BF16
                                 ? lora::launch_3d_bf16_lanes(p, p.steps_per_launch, d_in, d_out, begin, end, static_cast<hipStream_t>(stream))
                                 : lora::launch_3d_lanes(p, p.steps_per_launch, static_cast<const double *>(d_in),
                                                         static_cast<double *>(d_out), begin, end, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) {
            lora::set_last_error("fused register-resident 3D kernel launch", e);
            return LORA_EHIP;
        }
        return LORA_OK;
    }
    if (p.ndim == 3 && p.steps_per_launch == 3)  // an even global step: level 1 has the zero halo, level 2 the source's
        return lora::step3_natural(p, d_in, d_out, d_in, 0, begin, end, stream);
    if (p.ndim != 1) return lora_plan_step2_region(plan, d_in, d_out, begin, end, stream);
    if (int rc = lora::check_buffers(d_in, d_out)) return rc;
    if (d_in == d_out || begin < 0 || end > p.dims[0] || begin > end || (begin & 1)) return LORA_EINVAL;
    const hipError_t e = lora::launch_1d_fused(p, static_cast<const double *>(d_in), static_cast<double *>(d_out), begin,
                                               end, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        lora::set_last_error("fused 1D kernel launch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

int lora_plan_stepk(lora_plan *plan, const void *d_in, void *d_out, void *stream) {
    if (!plan) return LORA_EINVAL;
    return lora_plan_stepk_region(plan, d_in, d_out, 0, plan->p.dims[0], stream);
}

// `napps` applications in one launch where the plan's kernel family has that depth: 1, the plan's own depth, or one of
// the shallower depths its runs use for their tails (2D workgroup-row kernel: 4 and 2 under a six-application plan; 1D:
// the powers of two below the plan's depth; otherwise 2).  What the slab / block drivers issue for the tail of a run.
int lora_plan_stepn_region(lora_plan *plan, int napps, const void *d_in, void *d_out, int begin, int end, void *stream) {
    if (!plan || napps < 1) return LORA_EINVAL;
    Plan &p = plan->p;
    if (napps == 1) return lora_plan_step_region(plan, d_in, d_out, begin, end, stream);
    if (napps == p.steps_per_launch) return lora_plan_stepk_region(plan, d_in, d_out, begin, end, stream);
    if (napps > p.steps_per_launch) return LORA_EUNSUPPORTED;
    const bool wg_tail = p.ndim == 2 && p.wg_active && (napps == 4 || napps == 2);
    const bool d1_tail = p.ndim == 1 && (napps & (napps - 1)) == 0;
    if (wg_tail || d1_tail) {
        const int depth = p.steps_per_launch;
        p.steps_per_launch = napps;
        const int rc = lora_plan_stepk_region(plan, d_in, d_out, begin, end, stream);
        p.steps_per_launch = depth;
        return rc;
    }
    if (napps == 2) return lora_plan_step2_region(plan, d_in, d_out, begin, end, stream);
    return LORA_EUNSUPPORTED;
}

// The launches of one run, in order, on `stream` (also what gets captured into a hipGraph).
// How a run of `times` steps is cut into launches: nk launches of the plan's K applications, n2 shallower fused launches
// (1D and 2D; their depths in `tail`), the rest single sweeps.  The fused launches must leave the data in buffer 0.
struct FusedSchedule {
    int nk = 0, n2 = 0;
    bool scratch = false;  // odd number of fused launches: the last two hops go through the plan's scratch grid
    int tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // depths of the n2 launches after the nk full-depth ones
};

static void drop_graph(lora_plan *plan);

// the scratch grid this plan would use on the current device is there already
static bool scratch_ready(const lora_plan *plan) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    return plan->scratch && plan->scratch_bytes == lora_plan_padded_bytes(plan) && plan->scratch_device == dev;
}

static bool ensure_scratch(lora_plan *plan) {
    const size_t bytes = lora_plan_padded_bytes(plan);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (scratch_ready(plan)) return true;
    drop_graph(plan);  // a captured run holds the old grid's address
    if (plan->scratch) (void) hipFree(plan->scratch);
    plan->scratch = nullptr;
    if (hipMalloc(&plan->scratch, bytes) != hipSuccess) {
        (void) hipGetLastError();
        plan->scratch = nullptr;
        return false;
    }
    // cells the kernels never write (pads beyond the halo ring do not exist; the ring itself is copied per run)
    if (hipMemset(plan->scratch, 0, bytes) != hipSuccess) (void) hipGetLastError();
    plan->scratch_bytes = bytes;
    plan->scratch_device = dev;
    return true;
}

static FusedSchedule fused_schedule(lora_plan *plan, int times, bool can_fuse, bool allocate = true) {
    const Plan &p = plan->p;
    FusedSchedule fs;
    const int K = p.steps_per_launch;
    if (!can_fuse || K < 2) return fs;
    fs.nk = times / K;
    fs.n2 = 0;
    if (p.ndim == 2 || (p.ndim == 3 && K == 4)) {
        // 2D (and 3D with four applications per launch): what the full-depth launches leave is covered by one launch of four and / or one of two applications (the
        // row-streaming kernel), so that at most one single sweep remains: 100 sweeps at depth 6 = 16 x 6 + 4
        int r = times - K * fs.nk;
        if (K == 6 && r >= 2 && r < 4 && fs.nk >= 1) {
            // 6 + 2 as 4 + 4: a two-application launch moves the whole grid for two sweeps (star2d1r 16384^2, us per
            // launch of the workgroup-row kernel at 6 / 4 / 2 applications: 1194 / 944 / 885)
            fs.nk -= 1;
            fs.tail[fs.n2++] = 4;
            fs.tail[fs.n2++] = 4;
            r -= 2;
        }
        for (int d = 4; d >= 2; d -= 2)
            if (d < K && r >= d) {
                fs.tail[fs.n2++] = d;
                r -= d;
            }
        const int n = fs.nk + fs.n2;
        if (n % 2 == 0) return fs;
        if (n >= 3 && p.use_scratch != 0 && (allocate ? ensure_scratch(plan) : scratch_ready(plan))) {
            fs.scratch = true;
            return fs;
        }
        // no scratch grid: an even number of launches -- one full-depth launch becomes two shallower ones (6 = 4 + 2,
        // 4 = 2 + 2), or the last launch becomes single sweeps
        if (fs.nk >= 1 && K >= 4) {
            fs.nk -= 1;
            for (int q = fs.n2 - 1; q >= 0; --q) fs.tail[q + 2] = fs.tail[q];
            fs.tail[0] = K == 6 ? 4 : 2;
            fs.tail[1] = 2;
            fs.n2 += 2;
        } else if (fs.n2 > 0 && fs.tail[fs.n2 - 1] >= 4) {
            fs.tail[fs.n2 - 1] = 2;  // a four-application launch as two of two
            fs.tail[fs.n2++] = 2;
        } else if (fs.n2 > 0) {
            fs.n2 -= 1;
        } else {
            fs.nk -= 1;
        }
        return fs;
    }
    if (p.ndim == 1) {
        // 1D: what the full-depth launches leave is covered by shallower launches (K / 2, K / 4, ... 2 applications), so
        // that at most one single sweep remains: 100 sweeps at depth 32 = 32 + 32 + 32 + 4
        int r = times - K * fs.nk;
        for (int d = K / 2; d >= 2; d /= 2)
            if (r >= d) {
                fs.tail[fs.n2++] = d;
                r -= d;
            }
        const int n1 = fs.nk + fs.n2;
        if (n1 % 2 == 0) return fs;
        if (n1 >= 3 && p.use_scratch != 0 && (allocate ? ensure_scratch(plan) : scratch_ready(plan))) {
            fs.scratch = true;
            return fs;
        }
        // no scratch grid: an even number of launches -- the last launch becomes two of half its depth, or two single sweeps
        if (fs.n2 > 0) {
            const int d = fs.tail[fs.n2 - 1];
            if (d >= 4) {
                fs.tail[fs.n2 - 1] = d / 2;
                fs.tail[fs.n2++] = d / 2;
            } else {
                fs.n2 -= 1;
            }
        } else if (K >= 4) {
            fs.nk -= 1;
            fs.tail[fs.n2++] = K / 2;
            fs.tail[fs.n2++] = K / 2;
        } else {
            fs.nk -= 1;
        }
        return fs;
    }
    const int n = fs.nk + fs.n2;
    if (n % 2 == 0) return fs;
    if (n >= 3 && p.use_scratch != 0 && (allocate ? ensure_scratch(plan) : scratch_ready(plan))) {
        fs.scratch = true;
        return fs;
    }
    fs.nk -= 1;  // no scratch grid: an even number of launches instead
    return fs;
}

// Does lora_plan_run cut this plan's runs into fused launches at all, and is it the three-application 3D schedule (which
// runs on the reference's own buffer state: no scratch grid)?  One place, used by run_launches and by the pre-capture
// allocation in lora_plan_run.
static bool run_can_fuse(const Plan &p) {
    const int K = p.steps_per_launch;
    return p.boundary != LORA_BC_PERIODIC && K >= 2 &&
           (!p.generic || (p.ndim == 2 && p.stream2 && p.boundary == LORA_BC_REFERENCE) || (p.ndim == 3 && p.lanes3_active)) &&
           ((p.ndim == 2 && p.variant == LORA_VARIANT_DIRECT) || p.ndim == 3 || p.ndim == 1);
}
static bool run_is_natural3(const Plan &p) { return run_can_fuse(p) && p.ndim == 3 && p.steps_per_launch == 3; }

struct RunMarks {  // lora_plan_run_profiled: events around the fused and the single-sweep segment
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // start | K-launches | 2-launches | singles
    int fused_launches = 0, two_launches = 0, single_launches = 0;
};

// Test support (no device): replay, on the host, the decode the register-resident 3D kernels run for a launch cut into spans
// (team = 0) or team spans (team = 1) -- csrc/spans.h -- and count how often each (tile, plane) pair is swept.
// cover[(ty * tiles_x + tx) * depth + z] += 1 per workgroup segment that holds the pair; *workgroups = the launch's size,
// *max_steps = the busiest workgroup's steps (S per segment + its planes).  Every count must be 1.
int lora_debug_span_cover(int tiles_x, int tiles_y, int depth, int S, int slots, int team, int *cover, int *workgroups, int *max_steps) {
    if (tiles_x < 1 || tiles_y < 1 || depth < 1 || S < 1 || slots < 1 || !cover) return LORA_EINVAL;
    lora::Spans sp{};
    long wgs;
    if (team == 2) {
        // team spans as the bf16 kernel launches them: column lines with weighted rim rows, the rim columns cut finer
        if (tiles_x > slots) return LORA_EUNSUPPORTED;
        lora::Spans sp2{};
        long ni, nr;
        lora::team_pieces(tiles_x, slots, true, &ni, &nr);
        const bool weigh = tiles_x >= 3;
        const long ti = ni > 0 ? lora::spans_setup_column(sp, tiles_y, depth, S, ni, weigh ? 19 : 1, weigh ? 18 : 1) : 0;
        const long tr = nr > 0 ? lora::spans_setup_column(sp2, tiles_y, depth, S, nr, 1, 1) : 0;
        if (ti <= 0 || tr < ti) return LORA_EUNSUPPORTED;
        const long team_nr = tiles_x >= 3 ? tr : ti;
        wgs = tiles_x >= 3 ? ti * (tiles_x - 2) + 2 * tr : ti * tiles_x;
        int busiest2 = 0;
        for (long lin = 0; lin < wgs; ++lin) {
            long j;
            int col;
            if (lin < (long) tiles_x * ti) {
                j = lin / tiles_x;
                col = (int) (lin - j * tiles_x);
            } else {
                const long r = lin - (long) tiles_x * ti;
                j = ti + (r >> 1);
                col = (r & 1) ? tiles_x - 1 : 0;
            }
            const bool rim_col = team_nr != ti && (col == 0 || col == tiles_x - 1);
            const lora::Spans &t = rim_col ? sp2 : sp;
            unsigned v0, v1;
            lora::span_range(t, (int) j, v0, v1);
            int steps = 0;
            for (bool more = true; more;) {
                int t0, ty, z0, zc;
                const bool any = lora::span_next(t, S, v0, v1, 1, tiles_y, t0, ty, z0, zc);
                ty = lora::column_line_row(ty, tiles_y);
                more = v0 < v1;
                if (!any) continue;
                if (ty < 0 || ty >= tiles_y || z0 < 0 || z0 + zc > depth) return LORA_EHIP;  // (out of range: a bug)
                for (int z = z0; z < z0 + zc; ++z) cover[((long) ty * tiles_x + col) * depth + z] += 1;
                steps += S + zc;
            }
            busiest2 = std::max(busiest2, steps);
        }
        if (workgroups) *workgroups = (int) wgs;
        if (max_steps) *max_steps = busiest2;
        return LORA_OK;
    }
    if (team) {
        if (tiles_x > slots) return LORA_EUNSUPPORTED;
        wgs = lora::spans_setup(sp, 1, tiles_y, depth, S, slots / tiles_x, 1, 1) * tiles_x;
    } else {
        wgs = lora::spans_setup(sp, tiles_x, tiles_y, depth, S, slots, 10, 9);
    }
    if (wgs <= 0) return LORA_EUNSUPPORTED;
    int busiest = 0;
    for (long lin = 0; lin < wgs; ++lin) {
        unsigned v0, v1;
        lora::span_range(sp, (int) (team ? lin / tiles_x : lin), v0, v1);
        int steps = 0;
        for (bool more = true; more;) {
            int tx, ty, z0, zc;
            bool any;
            if (team) {
                int t0;
                any = lora::span_next(sp, S, v0, v1, 1, tiles_y, t0, ty, z0, zc);
                tx = (int) (lin % tiles_x);
            } else {
                any = lora::span_next(sp, S, v0, v1, tiles_x, tiles_y, tx, ty, z0, zc);
            }
            more = v0 < v1;
            if (!any) continue;
            if (tx < 0 || tx >= tiles_x || ty < 0 || ty >= tiles_y || z0 < 0 || z0 + zc > depth) return LORA_EHIP;  // (out of range: a bug)
            for (int z = z0; z < z0 + zc; ++z) cover[((long) ty * tiles_x + tx) * depth + z] += 1;
            steps += S + zc;
        }
        busiest = std::max(busiest, steps);
    }
    if (workgroups) *workgroups = (int) wgs;
    if (max_steps) *max_steps = busiest;
    return LORA_OK;
}

// Two ranges of planes / rows in one call: the two end regions a slab or block driver sweeps behind its deferred wait.
// The register-resident 3D kernels take both in ONE launch (an end region of 4 planes is 13 steps of pipeline whatever it
// computes: 32 us each at 64 x 512^2, tools/region_split_time.py); every other kernel family gets two launches.
int lora_plan_stepn_region2(lora_plan *plan, int napps, const void *d_in, void *d_out, int begin0, int end0, int begin1, int end1,
                            void *stream) {
    if (!plan || napps < 1) return LORA_EINVAL;
    Plan &p = plan->p;
    if (end0 <= begin0) return end1 <= begin1 ? LORA_OK : lora_plan_stepn_region(plan, napps, d_in, d_out, begin1, end1, stream);
    if (end1 <= begin1) return lora_plan_stepn_region(plan, napps, d_in, d_out, begin0, end0, stream);
    const bool lanes = p.ndim == 3 && p.lanes3_active && (napps == 4 || napps == 2) && napps <= p.steps_per_launch;
    const bool apart = end0 <= begin1 || end1 <= begin0;
    if (lanes && apart) {
        if (int rc = lora::check_buffers(d_in, d_out)) return rc;
        if (d_in == d_out || begin0 < 0 || end0 > p.dims[0] || begin1 < 0 || end1 > p.dims[0]) return LORA_EINVAL;
        hipStream_t s = static_cast<hipStream_t>(stream);
        const hipError_t e = p.dtype == LORA_BF16
                                 ? lora::launch_3d_bf16_lanes(p, napps, d_in, d_out, begin0, end0, s, begin1, end1)
                                 : lora::launch_3d_lanes(p, napps, static_cast<const double *>(d_in), static_cast<double *>(d_out), begin0,
                                                         end0, s, begin1, end1);
        if (e != hipSuccess) {
            lora::set_last_error("fused register-resident 3D kernel launch (two ranges)", e);
            return LORA_EHIP;
        }
        return LORA_OK;
    }
    if (int rc = lora_plan_stepn_region(plan, napps, d_in, d_out, begin0, end0, stream)) return rc;
    return lora_plan_stepn_region(plan, napps, d_in, d_out, begin1, end1, stream);
}

// ---- the periodic option in fused launches: the torus by ghost zones ---------------------------------------------
// A fused launch cannot wrap the halo of its intermediate levels, so rounds 1 - 3 ran periodic grids one sweep at a time
// behind a halo wrap each.  What the slab drivers do for a cut serves for the torus too: extend the grid by a ghost zone
// of radius x applications-per-launch cells on EVERY side, fill it with the periodic images of the interior, and run the
// ordinary fused kernels on the extended grid -- ghost cells are plain interior cells of that problem, computed by the same
// arithmetic on the same values as their originals, so after a launch of K applications everything but the outermost
// radius x K cells is right, the true interior included; wrap the ring again, launch again.  Costs two more buffers and,
// per launch, a wrap of the ring (surface work) and the sweeps' share of the ghost cells (16384^2, K = 6: 0.4 %).
// Bit-identical to wrap + single sweep wherever the fused kernels are bit-identical to single sweeps.
static int torus_radius(int nd) { return nd == 1 ? 4 : (nd == 2 ? 3 : 1); }
static const int *torus_pads(int nd) {
    static const int h1[1] = {4}, h2[2] = {4, 4}, h3[3] = {1, 2, 4};  // (kernels_halo.hip: launch_halo)
    return nd == 1 ? h1 : (nd == 2 ? h2 : h3);
}

static void torus_drop(lora_plan *plan) {
    if (plan->torus) lora_plan_destroy(plan->torus);
    plan->torus = nullptr;
    for (void *&b : plan->torus_buf) {
        if (b) (void) hipFree(b);
        b = nullptr;
    }
    plan->torus_bytes = 0;
    plan->torus_tried = false;
}

// the extended plan and its buffers, built on first need; nullptr: this plan's periodic runs stay single sweeps
static lora_plan *torus_prepare(lora_plan *plan) {
    Plan &p = plan->p;
    if (p.boundary != LORA_BC_PERIODIC || p.torus == 0 || p.steps_per_launch_req == 1 || p.variant != LORA_VARIANT_DIRECT) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void) hipGetLastError();
        return nullptr;
    }
    if (plan->torus_tried && plan->torus_epoch == p.epoch && plan->torus_device == dev) return plan->torus;
    if (plan->capturing) return nullptr;  // (no allocation inside a stream capture; lora_plan_run prepares before it)
    torus_drop(plan);
    plan->torus_tried = true;
    plan->torus_epoch = p.epoch;
    plan->torus_device = dev;
    const int nd = p.ndim, r = torus_radius(nd), kmax = nd == 1 ? 32 : (nd == 2 ? 6 : 4);
    const int *pad = torus_pads(nd);
    int ext[3] = {0, 0, 0};
    for (int d = 0; d < nd; ++d) {
        int g = r * kmax;
        if (d == nd - 1 && p.dtype == LORA_BF16) g = (g + 3) / 4 * 4;  // rows stay whole 8-byte units, a multiple of 8 cells
        if (g + pad[d] > p.dims[d]) return nullptr;  // an image of an image: grids this small keep the single sweeps
        plan->torus_ghost[d] = g;
        ext[d] = p.dims[d] + 2 * g;
    }
    lora_plan *tp = nullptr;
    if (lora_plan_create(&tp, p.shape, p.dtype, ext, nullptr) != LORA_OK) return nullptr;
    bool ok = lora_plan_set_boundary(tp, LORA_BC_REFERENCE) == LORA_OK && lora_plan_set_weights(tp, p.w, p.ntaps) == LORA_OK;
    if (ok && p.steps_per_launch_req > 1) (void) lora_plan_set_option(tp, "steps_per_launch", p.steps_per_launch_req);
    ok = ok && tp->p.steps_per_launch >= 2 && tp->p.steps_per_launch <= kmax;
    const size_t bytes = ok ? lora_plan_padded_bytes(tp) : 0;
    for (int i = 0; ok && i < 2; ++i)
        if (hipMalloc(&plan->torus_buf[i], bytes) != hipSuccess) {
            (void) hipGetLastError();
            plan->torus_buf[i] = nullptr;
            ok = false;
        }
    if (!ok) {
        lora_plan_destroy(tp);
        for (void *&b : plan->torus_buf) {
            if (b) (void) hipFree(b);
            b = nullptr;
        }
        return nullptr;
    }
    plan->torus = tp;
    plan->torus_bytes = bytes;
    return tp;
}

constexpr int kTorusNotTaken = -1000;

static int run_torus(lora_plan *plan, void *d_buf0, void *d_buf1, int times, hipStream_t s, RunMarks *marks) {
    if (times < 2) return kTorusNotTaken;
    lora_plan *tp = torus_prepare(plan);
    if (!tp) return kTorusNotTaken;
    const Plan &p = plan->p;
    const int nd = p.ndim, K = tp->p.steps_per_launch;
    const int *pad = torus_pads(nd);
    int ring[3] = {0, 0, 0};
    for (int d = 0; d < nd; ++d) ring[d] = pad[d] + plan->torus_ghost[d];
    void *buf[2] = {d_buf0, d_buf1};
    void *E[2] = {plan->torus_buf[0], plan->torus_buf[1]};
    auto hip = [&](hipError_t e, const char *what) -> int {
        if (e == hipSuccess) return LORA_OK;
        lora::set_last_error(what, e);
        return LORA_EHIP;
    };
    // level 0: the interior into the extended grid's middle, its images into everything around it
    if (int rc = hip(lora::launch_copy_interior(p.dtype, nd, p.dims, E[0], ring, buf[0], pad, s), "torus: copy in")) return rc;
    if (int rc = hip(lora::launch_ring_wrap(p.dtype, nd, p.dims, ring, E[0], s), "torus: wrap")) return rc;
    int cur = 0, left = times;
    while (left > 0) {
        int d = 1;
        if (left >= K)
            d = K;
        else if (nd == 1)
            while (2 * d <= left && 2 * d <= K) d *= 2;
        else if (nd == 2 && K == 6 && left >= 4)
            d = 4;
        else if (left >= 2)
            d = 2;
        int rc = lora_plan_stepn_region(tp, d, E[cur], E[1 - cur], 0, tp->p.dims[0], s);
        while (rc == LORA_EUNSUPPORTED && d > 1) {  // (a depth this plan's kernels do not have: the next one down)
            d = d > 2 ? 2 : 1;
            rc = lora_plan_stepn_region(tp, d, E[cur], E[1 - cur], 0, tp->p.dims[0], s);
        }
        if (rc != LORA_OK) return rc;
        if (int rc2 = hip(lora::launch_ring_wrap(p.dtype, nd, p.dims, ring, E[1 - cur], s), "torus: wrap")) return rc2;
        if (marks) {
            if (d == K)
                ++marks->fused_launches;
            else if (d > 1)
                ++marks->two_launches;
            else
                ++marks->single_launches;
        }
        cur = 1 - cur;
        left -= d;
    }
    if (marks) {
        (void) hipEventRecord(marks->ev[1], s);
        (void) hipEventRecord(marks->ev[2], s);
    }
    // the result: the extended grid's middle back into the caller's buffer, and that buffer's halo = its periodic images
    if (int rc = hip(lora::launch_copy_interior(p.dtype, nd, p.dims, buf[times % 2], pad, E[cur], ring, s), "torus: copy out")) return rc;
    const int rc = hip(lora::launch_halo(p, buf[times % 2], nullptr, lora::HALO_WRAP, s), "periodic halo");
    if (marks) (void) hipEventRecord(marks->ev[3], s);
    return rc;
}

static int run_launches(lora_plan *plan, void *d_buf0, void *d_buf1, int times, void *stream, RunMarks *marks = nullptr) {
    Plan &p = plan->p;
    void *buf[2] = {d_buf0, d_buf1};
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto mark = [&](int k) {
        if (marks) (void) hipEventRecord(marks->ev[k], s);
    };
    mark(0);
    if (times == 0) {
        mark(1);
        mark(2);
        mark(3);
        return LORA_OK;
    }
    auto halo = [&](void *dst, const void *src, int mode, const char *what) -> int {
        const hipError_t e = lora::launch_halo(p, dst, src, mode, s);
        if (e != hipSuccess) {
            lora::set_last_error(what, e);
            return LORA_EHIP;
        }
        return LORA_OK;
    };
    if (p.boundary != LORA_BC_REFERENCE) {
        if (int rc = lora::check_buffers(d_buf0, d_buf1)) return rc;
    }

    if (p.boundary == LORA_BC_PERIODIC) {
        // torus: fused launches on a ghost-extended grid where that applies (run_torus); else refresh the source's halo
        // from the opposite interior edges before every sweep, and once more at the end so that the result is a
        // consistent periodic array
        {
            const int rc = run_torus(plan, d_buf0, d_buf1, times, s, marks);
            if (rc != kTorusNotTaken) return rc;
        }
        mark(1);
        mark(2);
        for (int i = 0; i < times; ++i) {
            if (int rc = halo(buf[i % 2], nullptr, lora::HALO_WRAP, "periodic halo")) return rc;
            if (int rc = lora_plan_step(plan, buf[i % 2], buf[(i + 1) % 2], stream)) return rc;
        }
        if (marks) marks->single_launches = times;
        const int rc = halo(buf[times % 2], nullptr, lora::HALO_WRAP, "periodic halo");
        mark(3);
        return rc;
    }

    const bool dirichlet = p.boundary == LORA_BC_DIRICHLET;
    if (dirichlet) {
        // fixed halo: both buffers carry the caller's halo for the whole run
        if (int rc = halo(buf[1], buf[0], lora::HALO_COPY, "halo copy")) return rc;
    }
    int done = 0;
    const int K = p.steps_per_launch;  // applications per fused launch: 2 (2D, 3D) or 2 / 4 / 8 (1D)
    const bool can_fuse = run_can_fuse(p);
    FusedSchedule fs;
    const bool natural3 = run_is_natural3(p);
    if (natural3) {
        // Three applications per launch (3D fp64): launch k covers global steps 3 k + 1 .. 3 k + 3, reading buffer
        // k mod 2 and writing the other one -- exactly where the step-by-step driver has these levels, so the launches
        // run on the reference's own buffer state (buffer 0: the caller's halo, buffer 1: zeros; both the caller's
        // under the Dirichlet option): no halo copies, no parity constraint on the number of launches, no scratch grid.
        // The kernel is told the parity of its first step (which of its inner levels sees which halo).
        // What three-application launches leave (1 or 2 sweeps) would be single sweeps at a third of the rate.  When the
        // launches before them end at an even step in buffer 0, two of them are traded for TWO-application launches of
        // the same kernel instead: 3 a + 1 = 3 (a - 1) + 2 + 2, 3 a + 2 = 3 (a - 2) + 4 x 2 (50 sweeps = 14 x 3 + 4 x 2).
        // Those run like every other two-application launch: both buffers carry the caller's halo meanwhile.
        int nk = times / 3, n2 = 0;
        const int r = times - 3 * nk;
        if (r == 1 && nk >= 1 && (nk - 1) % 2 == 0) {
            nk -= 1;
            n2 = 2;
        } else if (r == 2 && nk >= 2 && (nk - 2) % 2 == 0) {
            nk -= 2;
            n2 = 4;
        }
        if (nk + n2 > 0) {
            if (int rc = lora::check_buffers(d_buf0, d_buf1)) return rc;
            if (!dirichlet)
                if (int rc = halo(buf[1], nullptr, lora::HALO_ZERO, "halo reset")) return rc;
            for (int k = 0; k < nk; ++k)
                if (int rc = lora::step3_natural(p, buf[k % 2], buf[(k + 1) % 2], buf[0], k & 1, 0, p.dims[0], stream))
                    return rc;
            mark(1);
            if (n2 > 0) {
                if (!dirichlet)
                    if (int rc = halo(buf[1], buf[0], lora::HALO_COPY, "halo copy")) return rc;
                for (int k = 0; k < n2; ++k)
                    if (int rc = lora_plan_step2(plan, buf[k % 2], buf[(k + 1) % 2], stream)) return rc;
                if (!dirichlet)
                    if (int rc = halo(buf[1], nullptr, lora::HALO_ZERO, "halo reset")) return rc;
            }
            done = 3 * nk + 2 * n2;
            if (marks) marks->fused_launches = nk;
            if (marks) marks->two_launches = n2;
        } else {
            mark(1);
        }
    } else {
        fs = fused_schedule(plan, times, can_fuse, /*allocate=*/!plan->capturing);
    }
    if (fs.nk + fs.n2 > 0) {
        // Temporal fusion.  A fused launch reads a buffer whose halo is the level-0 halo and writes another one, so
        // while fused launches run every physical buffer carries buffer 0's halo; the data must end in buffer 0, after
        // which (reference boundary) buffer 1's halo is put back to 0 and the remaining steps are single sweeps -- the
        // result and its halo end up exactly where the step-by-step driver leaves them.  An EVEN number of launches
        // ping-pongs there by itself; an odd number (>= 3) makes its last two hops through a scratch grid owned by the
        // plan (.. -> buffer 1 -> scratch -> buffer 0), which costs memory -- one more grid of 288 GB -- instead of a
        // four-sweep launch replaced by two two-sweep ones (star2d1r 16384^2, 20 sweeps: 6.2 -> 5.5 ms).  With the
        // Dirichlet boundary all halos simply stay.
        if (int rc = lora::check_buffers(d_buf0, d_buf1)) return rc;
        const int n = fs.nk + fs.n2;
        void *scratch = fs.scratch ? plan->scratch : nullptr;
        if (!dirichlet)
            if (int rc = halo(buf[1], buf[0], lora::HALO_COPY, "halo copy")) return rc;
        if (scratch)
            if (int rc = halo(scratch, buf[0], lora::HALO_COPY, "halo copy")) return rc;
        for (int k = 0; k < n; ++k) {
            void *src = buf[k % 2], *dst = buf[(k + 1) % 2];
            if (scratch && k == n - 2) dst = scratch;  // k odd: buffer 1 -> scratch
            if (scratch && k == n - 1) {               // k even: scratch -> buffer 0
                src = scratch;
                dst = buf[0];
            }
            if (k == fs.nk) mark(1);
            int rc;
            if (p.ndim <= 2 && k >= fs.nk) {  // a shallower launch (1D: the same kernel at another depth; 2D: four or two)
                const int depth = p.steps_per_launch;
                p.steps_per_launch = fs.tail[k - fs.nk];
                rc = lora_plan_stepk(plan, src, dst, stream);
                p.steps_per_launch = depth;
            } else {
                rc = k < fs.nk ? lora_plan_stepk(plan, src, dst, stream) : lora_plan_step2(plan, src, dst, stream);
            }
            if (rc != LORA_OK) return rc;
        }
        if (fs.n2 == 0) mark(1);
        if (!dirichlet)
            if (int rc = halo(buf[1], nullptr, lora::HALO_ZERO, "halo reset")) return rc;
        done = K * fs.nk + 2 * fs.n2;
        if (p.ndim <= 2) {
            done = K * fs.nk;
            for (int q = 0; q < fs.n2; ++q) done += fs.tail[q];
        }
        if (marks) marks->fused_launches = fs.nk;
        if (marks) marks->two_launches = fs.n2;
    } else if (!natural3) {
        mark(1);
    }
    mark(2);
    for (int i = done; i < times; ++i) {  // 2d/gpu.cu:544-546
        const int rc = lora_plan_step(plan, buf[i % 2], buf[(i + 1) % 2], stream);
        if (rc != LORA_OK) return rc;
    }
    if (marks) marks->single_launches = times - done;
    mark(3);
    return LORA_OK;
}

static void drop_graph(lora_plan *plan) {
    if (plan->graph_exec) {
        (void) hipGraphExecDestroy(plan->graph_exec);
        plan->graph_exec = nullptr;
    }
    plan->graph_times = -1;
}

// 1D runs with the automatic launch depth fuse more applications per launch the longer the run is: 8 is the plan's
// own depth (what lora_plan_stepk and the slab drivers use), 16 / 32 pay from 32 / 64 sweeps on (2^20 points: 840 ->
// 999 -> 1066 GStencils/s per launch, 2^28: 1600 -> 1915 -> 1987; tools/k1d.sh) while short runs keep enough fused
// launches.  Scoped to one lora_plan_run / lora_plan_run_profiled call.
struct RunDepth1D {
    Plan &p;
    const int saved;
    RunDepth1D(Plan &plan, int times) : p(plan), saved(plan.steps_per_launch) {
        if (p.ndim == 1 && p.steps_per_launch_req == 0 && p.steps_per_launch == 8 && p.boundary != LORA_BC_PERIODIC)
            p.steps_per_launch = times >= 64 ? 32 : (times >= 32 ? 16 : 8);
    }
    ~RunDepth1D() { p.steps_per_launch = saved; }
};

// Everything lora_plan_run(times) would allocate on first need, now: the scratch grid of a schedule with an odd number of
// fused launches (a hipMalloc + hipMemset of one more padded grid -- usually a millisecond or two, but on a fresh device
// it has taken 60 ms, INSIDE whatever region the caller was timing: the "four times slower" runs of DESIGN section 7), the
// extended grid's plan and buffers of a periodic run.  Idempotent; lora_plan_run works without it.
int lora_plan_prepare_run(lora_plan *plan, int times) {
    if (!plan || times < 0) return LORA_EINVAL;
    Plan &p = plan->p;
    RunDepth1D depth(p, times);
    if (p.boundary == LORA_BC_PERIODIC) {
        if (times >= 2) (void) torus_prepare(plan);
        return LORA_OK;
    }
    if (run_can_fuse(p) && !run_is_natural3(p)) (void) fused_schedule(plan, times, true);
    return LORA_OK;
}

int lora_plan_run(lora_plan *plan, void *d_buf0, void *d_buf1, int times, void *stream) {
    if (!plan || times < 0) return LORA_EINVAL;
    Plan &p = plan->p;
    RunDepth1D depth(p, times);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Launch-bound runs (small grids, many steps: the reference's 1D size sweeps in ~2 us per step) are captured
    // once into a hipGraph and replayed; big grids gain nothing and are launched directly.  Capture needs a real
    // stream (not the legacy default one) and no capture already in progress; every kernel takes its taps as launch
    // arguments, so none does host-side work at launch.
    bool want = p.use_graph == 1 || (p.use_graph < 0 && times >= 16 && lora_plan_padded_bytes(plan) <= (64u << 20));
    if (want && s == nullptr) want = false;
    if (want) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
            (void) hipGetLastError();
            want = false;
        }
    }
    if (!want) return run_launches(plan, d_buf0, d_buf1, times, stream);

    if (!(plan->graph_exec && plan->graph_buf[0] == d_buf0 && plan->graph_buf[1] == d_buf1 &&
          plan->graph_times == times && plan->graph_epoch == p.epoch)) {
        drop_graph(plan);
        if (int rc = lora::check_buffers(d_buf0, d_buf1)) return rc;
        // a scratch grid, if this run's schedule wants one, is allocated BEFORE the capture (hipMalloc and the null-stream
        // memset do not belong inside one); while capturing, run_launches only uses a grid that is already there
        if (run_can_fuse(p) && !run_is_natural3(p)) (void) fused_schedule(plan, times, true);
        if (p.boundary == LORA_BC_PERIODIC && times >= 2) (void) torus_prepare(plan);  // (likewise: the extended grid's buffers)
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            (void) hipGetLastError();
            return run_launches(plan, d_buf0, d_buf1, times, stream);
        }
        plan->capturing = true;
        const int rc = run_launches(plan, d_buf0, d_buf1, times, stream);
        plan->capturing = false;
        e = hipStreamEndCapture(s, &graph);
        if (rc != LORA_OK || e != hipSuccess || !graph) {
            if (graph) (void) hipGraphDestroy(graph);
            (void) hipGetLastError();
            return rc != LORA_OK ? rc : run_launches(plan, d_buf0, d_buf1, times, stream);
        }
        e = hipGraphInstantiate(&plan->graph_exec, graph, nullptr, nullptr, 0);
        (void) hipGraphDestroy(graph);
        if (e != hipSuccess) {
            plan->graph_exec = nullptr;
            (void) hipGetLastError();
            return run_launches(plan, d_buf0, d_buf1, times, stream);
        }
        plan->graph_buf[0] = d_buf0;
        plan->graph_buf[1] = d_buf1;
        plan->graph_times = times;
        plan->graph_epoch = p.epoch;
    }
    const hipError_t e = hipGraphLaunch(plan->graph_exec, s);
    if (e != hipSuccess) {
        lora::set_last_error("hipGraphLaunch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

int lora_plan_run_profiled(lora_plan *plan, void *d_buf0, void *d_buf1, int times, void *stream,
                           lora_run_profile *profile) {
    if (!plan || times < 0 || !profile) return LORA_EINVAL;
    RunDepth1D depth(plan->p, times);
    RunMarks marks;
    struct Guard {
        RunMarks &m;
        ~Guard() {
            for (hipEvent_t e : m.ev)
                if (e) (void) hipEventDestroy(e);
        }
    } guard{marks};
    for (hipEvent_t &e : marks.ev) LORA_HIP_TRY(hipEventCreate(&e));
    const int rc = run_launches(plan, d_buf0, d_buf1, times, stream, &marks);
    if (rc != LORA_OK) return rc;
    LORA_HIP_TRY(hipEventSynchronize(marks.ev[3]));
    profile->fused_launches = marks.fused_launches;
    profile->apps_per_fused_launch = marks.fused_launches ? plan->p.steps_per_launch : 1;
    profile->two_launches = marks.two_launches;
    profile->single_launches = marks.single_launches;
    LORA_HIP_TRY(hipEventElapsedTime(&profile->fused_ms, marks.ev[0], marks.ev[1]));
    LORA_HIP_TRY(hipEventElapsedTime(&profile->two_ms, marks.ev[1], marks.ev[2]));
    LORA_HIP_TRY(hipEventElapsedTime(&profile->single_ms, marks.ev[2], marks.ev[3]));
    return LORA_OK;
}

// ---- group A: host-buffer operators ------------------------------------------------------------------

static const char *run_label(int shape) {
    switch (shape) {
        case LORA_1D1R:
            return "LoRAStencil(1D 1d1r): ";  // 1d/gpu_1r.cu:127
        case LORA_1D2R:
            return "LoRAStencil(1D 1d2r): ";  // 1d/gpu_2r.cu:129
        case LORA_STAR2D1R:
            return "LoRAStencil(2D star_2d1r): ";  // 2d/gpu.cu:549
        case LORA_STAR2D3R:
            return "LoRAStencil(2D star_2d3r): ";  // 2d/gpu.cu:474
        case LORA_BOX2D1R:
        case LORA_BOX2D3R:
            return "LoRAStencil(2D box_2d3r): ";  // 2d/gpu.cu:415 (one operator serves both box shapes)
        case LORA_STAR3D1R:
            return "LoRAStencil(3D star_3d1r): ";  // 3d/gpu_star.cu:185
        case LORA_BOX3D1R:
            return "LoRAStencil(3D box_3d1r): ";  // 3d/gpu_box.cu:216
        default:
            return "LoRAStencil(?): ";
    }
}

namespace {
struct DeviceBuffers {
    void *b[2] = {nullptr, nullptr};
    ~DeviceBuffers() {
        for (void *p : b)
            if (p) (void) hipFree(p);
    }
};
}  // namespace

int lora_run_host(int shape, const double *in, double *out, const double *params, int times, const int *dims,
                  int quiet, lora_run_info *info) {
    return lora_run_host_dtype(shape, LORA_F64, in, out, params, times, dims, quiet, info);
}

int lora_run_host_dtype(int shape, int dtype, const void *in, void *out, const double *params, int times,
                        const int *dims, int quiet, lora_run_info *info) {
    if (!in || !out || !dims || times < 0) return LORA_EINVAL;
    if (lora_device_count() <= 0) {
        g_last_error = "no HIP device visible";
        return LORA_ENODEVICE;
    }
    lora_plan *plan = nullptr;
    int rc = lora_plan_create(&plan, shape, dtype, dims, params);
    if (rc != LORA_OK) return rc;
    struct PlanGuard {
        lora_plan *p;
        ~PlanGuard() { lora_plan_destroy(p); }
    } guard{plan};

    using clock = std::chrono::steady_clock;
    const size_t count = lora_padded_count(shape, dims);
    const size_t esize = (dtype == LORA_BF16) ? 2 : sizeof(double);
    const size_t bytes = count * esize;
    DeviceBuffers dev;
    const auto t_total0 = clock::now();
    LORA_HIP_TRY(hipMalloc(&dev.b[0], bytes));
    LORA_HIP_TRY(hipMalloc(&dev.b[1], bytes));
    LORA_HIP_TRY(hipMemcpy(dev.b[0], in, bytes, hipMemcpyHostToDevice));  // whole padded input, halo included
    // warm-up (the reference has none): one sweep into buf1, which is then cleared again
    if (times > 0) {
        rc = lora_plan_step(plan, dev.b[0], dev.b[1], nullptr);
        if (rc != LORA_OK) return rc;
    }
    LORA_HIP_TRY(hipMemset(dev.b[1], 0, bytes));
    // a stream of our own: launch-bound runs are replayed from a hipGraph, which the legacy default stream cannot capture
    struct StreamGuard {
        hipStream_t s = nullptr;
        ~StreamGuard() {
            if (s) (void) hipStreamDestroy(s);
        }
    } sg;
    LORA_HIP_TRY(hipStreamCreateWithFlags(&sg.s, hipStreamNonBlocking));
    LORA_HIP_TRY(hipDeviceSynchronize());
    if (times >= 16 && bytes <= (64u << 20)) {
        // build (capture + instantiate) the graph outside the timed region, like the reference's setup work
        rc = lora_plan_run(plan, dev.b[0], dev.b[1], times, sg.s);
        if (rc != LORA_OK) return rc;
        LORA_HIP_TRY(hipStreamSynchronize(sg.s));
        LORA_HIP_TRY(hipMemcpy(dev.b[0], in, bytes, hipMemcpyHostToDevice));
        LORA_HIP_TRY(hipMemset(dev.b[1], 0, bytes));
    }
    // (set-up, like the reference's own allocations ahead of its timed loop: the scratch grid / extended grid this run's
    // schedule would otherwise allocate inside the timed region)
    (void) lora_plan_prepare_run(plan, times);
    LORA_HIP_TRY(hipDeviceSynchronize());

    const auto t0 = clock::now();
    rc = lora_plan_run(plan, dev.b[0], dev.b[1], times, sg.s);
    if (rc != LORA_OK) return rc;
    LORA_HIP_TRY(hipStreamSynchronize(sg.s));
    const auto t1 = clock::now();

    // 1D copies all but the last element (1d/gpu_1r.cu:134)
    const size_t copy_bytes = (plan->p.ndim == 1) ? bytes - esize : bytes;
    LORA_HIP_TRY(hipMemcpy(out, dev.b[times % 2], copy_bytes, hipMemcpyDeviceToHost));
    const auto t_total1 = clock::now();

    double points = 1.0;
    for (int d = 0; d < plan->p.ndim; ++d) points *= dims[d];
    const long long us = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
    const double secs = us / 1e6;
    const int F = lora_shape_gstencil_factor(shape);
    lora_run_info ri;
    ri.sweep_seconds = std::chrono::duration<double>(t1 - t0).count();
    ri.total_seconds = std::chrono::duration<double>(t_total1 - t_total0).count();
    ri.gstencils = points * times / ri.sweep_seconds / 1e9;
    ri.gstencils_refconv = ri.gstencils * F;
    ri.hbm_gbs = points * times * 2.0 * esize / ri.sweep_seconds / 1e9;
    ri.variant = plan->p.variant;
    ri.steps_per_launch = plan->p.steps_per_launch;
    lora::g_last_info = ri;
    if (info) *info = ri;
    if (!quiet) {
        // byte-compatible with the reference's three lines (2d/gpu.cu:549-553)
        std::printf("%s\n", run_label(shape));
        std::printf("Time = %lld[ms]\n",
                    (long long) std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
        std::printf("GStencil/s = %f\n", points * times * F / secs / 1e9);
        std::fflush(stdout);
    }
    return LORA_OK;
}

int lora_last_run_info(lora_run_info *info) {
    if (!info) return LORA_EINVAL;
    *info = lora::g_last_info;
    return LORA_OK;
}

int lora_gpu_1d1r(const double *in, double *out, const double *params, int times, int n) {
    const int dims[1] = {n};
    return lora_run_host(LORA_1D1R, in, out, params, times, dims, 0, nullptr);
}
int lora_gpu_1d2r(const double *in, double *out, const double *params, int times, int n) {
    const int dims[1] = {n};
    return lora_run_host(LORA_1D2R, in, out, params, times, dims, 0, nullptr);
}
int lora_gpu_star_2d1r(const double *in, double *out, const double *params, int times, int m, int n) {
    const int dims[2] = {m, n};
    return lora_run_host(LORA_STAR2D1R, in, out, params, times, dims, 0, nullptr);
}
int lora_gpu_star_2d3r(const double *in, double *out, const double *params, int times, int m, int n) {
    const int dims[2] = {m, n};
    return lora_run_host(LORA_STAR2D3R, in, out, params, times, dims, 0, nullptr);
}
int lora_gpu_box_2d3r(const double *in, double *out, const double *params, int times, int m, int n) {
    const int dims[2] = {m, n};
    return lora_run_host(LORA_BOX2D3R, in, out, params, times, dims, 0, nullptr);
}
int lora_gpu_box_3d1r(const double *in, double *out, const double *params, int times, int h, int m, int n) {
    const int dims[3] = {h, m, n};
    return lora_run_host(LORA_BOX3D1R, in, out, params, times, dims, 0, nullptr);
}
int lora_gpu_star_3d1r(const double *in, double *out, const double *params, int times, int h, int m, int n) {
    const int dims[3] = {h, m, n};
    return lora_run_host(LORA_STAR3D1R, in, out, params, times, dims, 0, nullptr);
}

}  // extern "C"

namespace lora {
const char *run_label(int shape) { return ::run_label(shape); }
}  // namespace lora
