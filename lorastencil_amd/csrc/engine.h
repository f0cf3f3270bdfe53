// engine.h -- internal interfaces of liblorastencil_hip (not installed; the public surface is
// include/lorastencil.h).
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>

#include "lorastencil.h"

namespace lora {

// ---- host helpers (weights.cpp) ------------------------------------------------------------
int shape_ndim(int shape);
int shape_ntaps(int shape);
int default_params(int shape, double *params);
int effective_weights(int shape, const double *params, double *weights);
int factorize_7x7(const double *params, double u[4][7], double v[4][7], double *residual_max);
int svd_7x7(const double *W, double u[7][7], double v[7][7], double sigma[7]);
int separable_27(const float *w27, float *cba9);
int separable_27d(const double *w27, double *cba9);  // the same in fp64 (3D fp64 plane-streaming kernel)  // exact rank-1 test of fp32 3x3x3 taps; cba = c(x), b(y), a(z)
int mfma_factors_27(const float *cba9, float *scale, float *cba9_normalised);  // bf16-exact normalised factors or 0

// ---- tap sets: which of the 49 / 27 taps a kernel instantiation evaluates --------------------
enum TapSet2D { TAPS2D_DIAMOND = 0, TAPS2D_STAR = 1, TAPS2D_BOX = 2 };
enum TapSet3D { TAPS3D_STAR = 0, TAPS3D_BOX = 1, TAPS3D_SEP = 2 };  // SEP: w = a (x) b (x) c exactly (bf16 path)

struct Taps9 {
    double w[9];
};
struct Taps49 {
    double w[49];
};
struct Taps27 {
    double w[27];
};

// Band factors of the low-rank MFMA formulation: out = sum_t (U_t X) V_t + sparse residual.
struct LowRank2D {
    int rank;         // 1..3 terms
    double u[3][7];   // vertical profile of term t
    double v[3][7];   // horizontal profile of term t
    int nresid;       // residual taps applied on the vector pipe
    int rdy[16], rdx[16];
    double rw[16];
};

// ---- plan -------------------------------------------------------------------------------------
struct Plan {
    int shape = 0, ndim = 0, dtype = LORA_F64;
    int dims[3] = {0, 0, 0};  // interior extents, outermost first
    int ntaps = 0;
    double w[49] = {0};  // taps applied per sweep
    int variant = LORA_VARIANT_DIRECT;
    int tapset = 0;
    // tuning knobs (lora_plan_set_option)
    int rows_per_thread = 8;  // 2D direct: output rows per lane (tile height = 4x this)
    int panel_width = 32;     // 2D: tile columns per L2 panel of the block->tile map
    int nt_store = 0;         // 2D: non-temporal output stores
    int fused_rows_req = 0;   // 0 = auto, else 6 / 8 / 10
    int fused_rows = 8;       // 2D fused: intermediate rows per wave (tile = 4x this - 6 output rows), resolved
    int cols_per_lane = 4;    // 3D bf16: 4 (512-byte row pieces per wave) or 8 (1 KiB)
    int separable = -1;       // 3D bf16: evaluate exactly-separable taps as x/y/z passes: -1 auto (= on), 0 off
    double sep64[9] = {0};    // fp64 factors c(x), b(y), a(z) of exactly separable 3D taps (plane-streaming kernel)
    int sep64_valid = 0;      // resolved: the fp64 taps are exactly separable and the option allows that form
    float sep[9] = {0};       // resolved factors c(x), b(y), a(z) when tapset == TAPS3D_SEP
    int mfma_split = 1;       // bf16 MFMA variant: intermediate as hi + lo bf16 halves (1, the contract) or one bf16 rounding (0)
    bool mfma3_valid = false; // bf16: the taps are scale * a (x) b (x) c with bf16-exact normalised factors (MFMA variant)
    float mfma3_scale = 0.0f, mfma3_abc[9] = {0};  // normalised c, b, a
    int ablate = 0;           // diagnostics only (2D fused, 3D bf16): 1 = skip stores, 2 = skip loads; results wrong
    int lds_dma = 0;          // 3D bf16: global_load_lds ring, two planes ahead (hand-counted vmcnt)
    int persistent = 0;       // 2D fused: persistent workgroups with register prefetch of the next tile
    int stream2 = 1;          // 2D fused: row-streaming kernel (kernels_2d_stream.hip, default) or the tile kernel (0)
    int stream_rows = 0;      // ... output rows per chunk (0 = auto: whole rounds of resident waves)
    int stream_depth = 4;     // ... input rows in flight per wave (2..6; at most 3 with four applications per launch)
    int stream_share = 0;     // ... one ring of whole rows per workgroup, a barrier per row (fewer, aligned L2 requests)
    int stream_prefetch = 0;  // ... K = 4: fetch the next level's LDS window while the current level computes (measured: no gain)
    int stream_sync = 1;      // ... s_barrier per 7 rows (1) / per row (2) keeps a workgroup's four strips in step
    int wg = -1;              // 2D fused launches through the workgroup-row kernel (kernels_2d_wg.hip): -1 = when the plan fuses six applications per launch (then also its four / two tails), 0 never, 1 always
    int wg_active = 0;        // resolved
    int wg_rows = 0;          // 2D workgroup-row kernel (kernels_2d_wg.hip): output rows per chunk (0 = auto: one round of resident workgroups)
    int wg_prio = 12;         // ... time-sliced wave priorities that share a CU evenly between its two workgroups: log2 of the slice in 10 ns ticks (0 = off)
    int wg_edge_pct = -1;     // ... how much shorter the chunks of the first / last strip are, in per cent of a step's cost (-1 = default)
    int z_chunk = 16;         // 3D: output planes streamed per workgroup
    int fused_pipeline = 0;   // 3D bf16 fused: 1 = level 2 one plane behind level 1, one barrier per plane (no gain measured)
    int stream3 = -1;         // 3D fp64 fused: plane-streaming kernel (kernels_3d_planes.hip: 2 or 3 applications per launch) always (1), never (0: the tile kernel, 2 applications), or by grid size (-1)
    int stream3_active = 0;   // resolved: fused launches go through the plane-streaming kernel
    int stream3_waves = 0;    // 3D plane-streaming kernel: waves per workgroup: 0 = automatic, 8 (one workgroup per CU) or 4 (two); output tiles of 8 x waves - 2 (K - 1) rows x 60 columns
    int stream3_pipe = 0;     // 3D plane-streaming kernel: 1 = every level consumes what was published one step earlier (one barrier per step, two buffers per level)
    int stream3_async = 0;    // 3D plane-streaming kernel: 1 = no workgroup barriers (neighbour-wave counters in LDS, private input rings)
    int stream3_slots = 0;    // 3D plane-streaming kernel: input plane slots of the LDS ring (0 = as many as fit)
    int lanes3 = -1;          // 3D fused launches through the register-resident kernels (kernels_3d_lanes.hip, fp64; kernels_3d_bf16_lanes.hip, bf16: four applications per launch): -1 by grid size, 0 never, 1 always (star / separable box taps, reference boundary)
    int lanes3_active = 0;    // resolved
    int fused_z_chunk = 0;    // 3D fused: output planes per workgroup (0 = auto: 32, shorter on small grids)
    int spans3 = -1;          // 3D register-resident kernels: cut the (tile, plane) line into equal pieces per CU (1), equal chunks per tile (0), by region depth (-1)
    int torus = 1;            // periodic boundary: runs in fused launches on a ghost-extended grid (1), single sweeps behind a wrap each (0)
    int steps_per_launch_req = 0;  // 0 = auto, 1, 2 (2D / 3D), 4 (2D row-streaming kernel), 2 / 4 / 8 (1D)
    int steps_per_launch = 1;      // resolved
    bool generic = false;  // odd innermost extent: rows are only 8-byte aligned, the tiled kernels do not apply
    int lowrank_valu = -1;    // 2D fused: low-rank evaluation on the vector pipe: -1 auto, 0 off, 1 on when the factors fit
    int fused_eval = 0;       // resolved: 0..2 = direct taps of `tapset`, 3 = low-rank diamond, 4..6 = low-rank pyramid forms, 7 = nested profiles
    double lowrank_rc = 0.0;  // weight of the diamond form's 8-point correction
    double nest_g[4] = {0, 0, 0, 0}, nest_a[4] = {0, 0, 0, 0};  // fused_eval 7: nested-profile form (rows_2d.h)
    bool lowrank_valid = false;
    LowRank2D lowrank{};
    std::string kernel_name;
    int boundary = LORA_BC_REFERENCE;  // what halo cells hold between sweeps (lora_plan_set_boundary)
    int use_scratch = -1; // odd numbers of fused launches route through a scratch grid owned by the plan: -1 / 1 yes, 0 no
    int use_graph = -1;   // -1 auto (small grids, many launches, non-default stream), 0 never, 1 whenever possible
    unsigned epoch = 0;   // bumped by every change of taps / options: invalidates a cached graph
};

void plan_refresh(Plan &p);  // re-derive tapset / low-rank factors / kernel name from w + options

// ---- kernel launchers (kernels_*.hip).  Interior index range [begin, end) of the outermost
// dimension; all return the launch status. ----------------------------------------------------
hipError_t launch_1d(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
// steps_per_launch (2 / 4 / 8) applications per launch, intermediate levels in LDS
hipError_t launch_1d_fused(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
const char *kernel_name_1d_fused(const Plan &p);
hipError_t launch_2d_direct(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
hipError_t launch_2d_mfma(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
// two applications per launch (intermediate level in LDS, its halo = 0); 2D direct taps only
hipError_t launch_2d_fused2(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
// halo cells of a padded array (any shape / element type): copy from src, zero, or periodic wrap from dst itself
enum HaloMode { HALO_COPY = 0, HALO_ZERO = 1, HALO_WRAP = 2 };
hipError_t launch_halo(const Plan &p, void *dst, const void *src, int mode, hipStream_t s);
// rows x cols doubles between two strided arrays (pack / unpack of a block decomposition's column ghost zones)
hipError_t launch_copy_block(double *dst, long dst_ld, const double *src, long src_ld, long rows, long cols, hipStream_t s);
// the torus by ghost zones (capi.cpp: run_torus): wrap of a ring of any width, interior-to-interior copies
hipError_t launch_ring_wrap(int dtype, int nd, const int *dims, const int *ring, void *ptr, hipStream_t s);
hipError_t launch_copy_interior(int dtype, int nd, const int *dims, void *dst, const int *pad_dst, const void *src, const int *pad_src,
                                hipStream_t s);
const char *kernel_name_2d_fused2(const Plan &p);
// the same two applications per launch, row-streaming form (wave-autonomous column strips)
// (K = 2 or 4 applications per launch)
hipError_t launch_2d_stream(const Plan &p, int K, const double *in, double *out, int begin, int end, hipStream_t s);
const char *kernel_name_2d_stream(const Plan &p);
int stream_rows_per_chunk(const Plan &p, int K, int rows_total, int strips);  // resolved chunk height of a launch
int stream_strip_width(int K);
// K = 6, 4 or 2 applications per launch: workgroup-wide rows, levels pipelined over two groups of waves (kernels_2d_wg.hip)
hipError_t launch_2d_wg(const Plan &p, int K, const double *in, double *out, int begin, int end, hipStream_t s);
const char *kernel_name_2d_wg(const Plan &p);
void prepare_2d_wg(const Plan &p);  // one-time host work of the plan's instantiations (no launch)
int wg_strip_width(int K);
hipError_t launch_3d(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
// two applications per launch, level 1 in LDS (fp64, reference boundary: level-1 halo = 0)
hipError_t launch_3d_fused2(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
const char *kernel_name_3d_fused2(const Plan &p);
hipError_t launch_3d_stream(const Plan &p, int K, const double *in, double *out, const double *halo_src, int parity,
                            int begin, int end, hipStream_t s);
const char *kernel_name_3d_stream(const Plan &p);
// K = 4 (or 2) applications per launch with the levels in registers (star / exactly separable box taps, fp64, any extents)
hipError_t launch_3d_lanes(const Plan &p, int K, const double *in, double *out, int begin, int end, hipStream_t s, int begin2 = 0, int end2 = 0);
const char *kernel_name_3d_lanes(const Plan &p);
bool prepare_3d_lanes(const Plan &p);
int stream3_slots(int K, int waves, int pipe, int requested);
int stream3_waves(const Plan &p, int K, int pipe);
// any size, any taps (odd innermost extents): one thread per point
hipError_t launch_2d_generic(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
hipError_t launch_3d_generic(const Plan &p, const double *in, double *out, int begin, int end, hipStream_t s);
const char *kernel_name_generic(const Plan &p);
hipError_t launch_3d_bf16(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s);
const char *kernel_name_3d_bf16(const Plan &p);
hipError_t launch_3d_bf16_fused2(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s);
const char *kernel_name_3d_bf16_fused2(const Plan &p);
// bf16, exactly separable box taps: K = 4 (or 2) applications per launch with the levels in registers (kernels_3d_bf16_lanes.hip)
hipError_t launch_3d_bf16_lanes(const Plan &p, int K, const void *in, void *out, int begin, int end, hipStream_t s, int begin2 = 0, int end2 = 0);
const char *kernel_name_3d_bf16_lanes(const Plan &p);
bool prepare_3d_bf16_lanes(const Plan &p);  // false: no workgroup of it fits a CU of the current device
// bf16 box, two applications per launch, in-plane passes on v_mfma_f32_16x16x32_bf16 (LORA_VARIANT_MFMA)
hipError_t launch_3d_bf16_mfma2(const Plan &p, const void *in, void *out, int begin, int end, hipStream_t s);
const char *kernel_name_3d_bf16_mfma2(const Plan &p);

const char *kernel_name_1d(const Plan &p);
const char *kernel_name_2d_direct(const Plan &p);
const char *kernel_name_2d_mfma(const Plan &p);
const char *kernel_name_3d(const Plan &p);
int region_granularity(const Plan &p);

void set_last_error(const char *what, hipError_t e);
void set_last_error_text(const char *text);
void set_last_run_info(const lora_run_info &info);  // what lora_last_run_info returns on this thread
const char *run_label(int shape);                  // the operator's first stdout line (e.g. 2d/gpu.cu:549)

}  // namespace lora

struct lora_plan {
    lora::Plan p;
    // hipGraph of the last lora_plan_run (launch-bound small grids): replayed while buffers / step count match
    hipGraphExec_t graph_exec = nullptr;
    void *graph_buf[2] = {nullptr, nullptr};
    int graph_times = -1;
    unsigned graph_epoch = 0;  // value of p.epoch the graph was captured at
    bool capturing = false;    // lora_plan_run is capturing its launches: no allocations meanwhile
    // one more padded grid, allocated on first need (lora_plan_run with an odd number of fused launches)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int scratch_device = -1;
    // the periodic option's fused runs (run_torus): a plan of the grid extended by a ghost zone on every side, its two
    // buffers, the ghost widths; rebuilt when the plan's taps / options change
    lora_plan *torus = nullptr;
    void *torus_buf[2] = {nullptr, nullptr};
    size_t torus_bytes = 0;
    int torus_device = -1;
    int torus_ghost[3] = {0, 0, 0};
    unsigned torus_epoch = 0;
    bool torus_tried = false;  // at torus_epoch: the answer was already "no" (grid too small, no fused kernel, no memory)
};
