/*
 * lorastencil.h -- C ABI of the MI355X-native low-rank stencil engine (liblorastencil_hip.so).
 *
 * This is the drop-in boundary for the hot path of zondie17/LoRAStencil: the stencil sweep and
 * its time-step driver behind `lorastencil_{1d,2d,3d} shape input_size time_size`.  Plain C,
 * plain pointers and sizes, no C++/torch types.  Every entry point names the reference
 * interface it replaces (file:line under /root/reference/src/).
 *
 * Three groups:
 *   A. host-buffer operators  -- one-to-one replacements of the reference's seven gpu_*()
 *      functions (same argument meaning, padded host arrays in and out, same stdout lines);
 *   B. device-resident plans  -- the same sweep on caller-owned device buffers and a caller
 *      stream (what bench.py, the multi-GPU slab driver and the CLIs are built on);
 *   C. host helpers on the path -- params tables, the low-rank factor precompute, the
 *      glibc-rand() fill of the reference harness.
 *
 * Error model: every function returns LORA_OK (0) or a negative LORA_E* code and never exits
 * the process (the reference prints and exit(1)/abort()s: 2d_utils.h:6-36).  The C++ shims in
 * lorastencil_ref_shims.h restore the reference's print-and-exit behaviour on top of this.
 */
#ifndef LORASTENCIL_H
#define LORASTENCIL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LORA_VERSION 100

/* ---- status codes ---------------------------------------------------------------------- */
#define LORA_OK 0
#define LORA_EINVAL (-1)       /* bad shape / null pointer / negative size               */
#define LORA_EUNSUPPORTED (-2) /* size/dtype/variant combination the kernels cannot take       */
#define LORA_EHIP (-3)         /* a HIP runtime call failed (see lora_last_error())        */
#define LORA_ENOMEM (-4)
#define LORA_ENODEVICE (-5) /* no HIP device visible: the engine has NO CPU fallback     */

/* ---- shapes: the CLI `shape` argument (README.md:37-48) ---------------------------------- */
typedef enum lora_shape {
    LORA_1D1R = 0,     /* 1d_utils.h:38-42 star_1d1r */
    LORA_1D2R = 1,     /* star_1d2r */
    LORA_STAR2D1R = 2, /* 2d_utils.h:38-44 star_2d1r */
    LORA_BOX2D1R = 3,  /* box_2d1r (served by the box2d3r operator, 2d/main.cu:276-279) */
    LORA_STAR2D3R = 4, /* star_2d3r */
    LORA_BOX2D3R = 5,  /* box_2d3r */
    LORA_STAR3D1R = 6, /* 3d_utils.h:39-42 star_3d1r */
    LORA_BOX3D1R = 7,  /* box_3d1r */
    LORA_NUM_SHAPES = 8
} lora_shape;

typedef enum lora_dtype {
    LORA_F64 = 0, /* the reference's only type (DATA_TYPE double, 2d_utils.h:1) */
    LORA_BF16 = 1 /* NEW (no reference counterpart): bf16 storage, fp32 accumulation in tap order, one
                     round-to-nearest-even per sweep; 3D shapes only, innermost extent a multiple of 8 */
} lora_dtype;

/* Kernel formulation of one sweep (B: lora_plan_set_variant). */
typedef enum lora_variant {
    LORA_VARIANT_AUTO = 0,   /* what the roofline evidence in DESIGN.md picked per shape           */
    LORA_VARIANT_DIRECT = 1, /* LDS-tiled direct taps on the fp64 vector FMA pipe                  */
    LORA_VARIANT_MFMA = 2    /* low-rank (U X) V products on v_mfma_f64_16x16x4 (2D shapes only)   */
} lora_variant;

/* What the halo cells hold between sweeps of the time-step driver (lora_plan_run and the host operators).
 * Only LORA_BC_REFERENCE is behaviour of the reference; the other two are SURVEY section 8f-3 options. */
typedef enum lora_boundary {
    LORA_BC_REFERENCE = 0, /* halo cells are never written: sweep i reads the halo of buffer i % 2 -- the caller's
                              values at even steps, the second buffer's (zeros) at odd ones            */
    LORA_BC_DIRICHLET = 1, /* halo cells keep the caller's input values at every time level                        */
    LORA_BC_PERIODIC = 2   /* the grid is a torus: before every sweep each halo cell is refreshed from the interior
                              cell it is a periodic image of (every extent must be >= its halo width)              */
} lora_boundary;

/* Number of taps in `params` / weights for a shape: 9 (1D), 49 (2D), 27 (3D). */
int lora_shape_ntaps(int shape);
/* Number of interior dimensions of a shape (1, 2 or 3), 0 for an unknown shape. */
int lora_shape_ndim(int shape);
/* Parse the CLI spelling ("star2d1r", ...) -> shape id, or LORA_EINVAL. */
int lora_shape_from_name(const char *name);
/* The spelling the reference prints in its INFO line ("star_2d1r", 2d/main.cu:5-10). */
const char *lora_shape_info_name(int shape);
/* Padded element count: 1D n+8; 2D (m+8)(n+8); 3D (h+2)(m+4)(n+8) (2d/main.cu:217-221). */
size_t lora_padded_count(int shape, const int *dims);
/* GStencil/s accounting factor F the reference multiplies by. */
int lora_shape_gstencil_factor(int shape);

const char *lora_strerror(int status);
/* Text of the last HIP failure on this thread ("" if none). */
const char *lora_last_error(void);
/* Number of visible HIP devices (0 if none; never fails). */
int lora_device_count(void);

/* ========================================================================================
 * A. Host-buffer operators: drop-in for the reference's gpu_*().
 *
 *   in, out : caller-owned PADDED host arrays (1D n+8; 2D (m+8)(n+8); 3D (h+2)(m+4)(n+8)),
 *             row-major; the whole padded `in` is read (halo values matter), the whole padded
 *             `out` is written (1D: all but the last element, 1d/gpu_1r.cu:134).
 *   params  : 9 / 49 / 27 doubles, row-major, centre at 4 / 24 / 13.  Honoured exactly as far as
 *             the reference honours them (ignored by star2d1r and star3d1r; box3d1r reads
 *             params[0..2]; box2d goes through the pyramid factoriser).
 *   times   : number of kernel applications; result = ping-pong buffer [times % 2].
 *   Side effect: prints the reference's three stdout lines (label, Time, GStencil/s).
 *   Device memory is allocated and freed inside the call.
 * ====================================================================================== */
int lora_gpu_1d1r(const double *in, double *out, const double *params, int times, int input_n);      /* gpu_1d1r      1d_utils.h:45, 1d/gpu_1r.cu:90   */
int lora_gpu_1d2r(const double *in, double *out, const double *params, int times, int input_n);      /* gpu_1d2r      1d_utils.h:47, 1d/gpu_2r.cu:91   */
int lora_gpu_star_2d1r(const double *in, double *out, const double *params, int times, int input_m,
                       int input_n);                                                                 /* gpu_star_2d1r 2d_utils.h:47, 2d/gpu.cu:484     */
int lora_gpu_star_2d3r(const double *in, double *out, const double *params, int times, int input_m,
                       int input_n);                                                                 /* gpu_star_2d3r 2d_utils.h:49, 2d/gpu.cu:426     */
int lora_gpu_box_2d3r(const double *in, double *out, const double *params, int times, int input_m,
                      int input_n);                                                                  /* gpu_box_2d3r  2d_utils.h:51, 2d/gpu.cu:276     */
int lora_gpu_box_3d1r(const double *in, double *out, const double *params, int times, int input_h,
                      int input_m, int input_n);                                                     /* gpu_box_3d1r  3d_utils.h:44, 3d/gpu_box.cu:143 */
int lora_gpu_star_3d1r(const double *in, double *out, const double *params, int times, int input_h,
                       int input_m, int input_n);                                                    /* gpu_star_3d1r 3d_utils.h:47, 3d/gpu_star.cu:136 */

/* Timing of the last host-buffer run, as the reference measures it (steady_clock around the
 * launch loop + one device sync, transfers excluded: 2d/gpu.cu:542-552), plus the transfer
 * inclusive time. */
typedef struct lora_run_info {
    double sweep_seconds;    /* launch loop + sync                      */
    double total_seconds;    /* H2D + sweep + D2H                       */
    double gstencils;        /* points*times/sweep_seconds/1e9 (F = 1)  */
    double gstencils_refconv; /* the same times the reference's factor F */
    double hbm_gbs;          /* algorithmic 2*sizeof(T) bytes per point */
    int variant;             /* lora_variant actually used              */
    int steps_per_launch;    /* 1 unless temporal fusion is active      */
} lora_run_info;

/* Generic form of group A (shape chosen at run time).  `quiet` != 0 suppresses the stdout
 * lines; `info` may be NULL. */
int lora_run_host(int shape, const double *in, double *out, const double *params, int times, const int *dims,
                  int quiet, lora_run_info *info);
/* The same for a given element type: `in` / `out` are padded host arrays of doubles (LORA_F64) or of bf16 bit
 * patterns in uint16_t (LORA_BF16). */
int lora_run_host_dtype(int shape, int dtype, const void *in, void *out, const double *params, int times,
                        const int *dims, int quiet, lora_run_info *info);
/* lora_run_info of the last group-A call on this thread (what the CLIs print after the reference's lines). */
int lora_last_run_info(lora_run_info *info);

/* ========================================================================================
 * B. Device-resident plans.
 *    A plan fixes shape, interior dims, weights and kernel variant; it owns no grid memory.
 *    Buffers are PADDED device arrays laid out like the host arrays of group A; `stream` is a
 *    hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous.
 *    Different plans may be used from different threads at the same time; one plan must not be (it caches the
 *    hipGraph of its last run and its options are plain fields).
 * ====================================================================================== */
typedef struct lora_plan lora_plan;

/* params -> effective weights exactly like the reference operator (incl. the low-rank factor
 * precompute for box2d).  params == NULL uses the reference harness's table for the shape. */
int lora_plan_create(lora_plan **plan, int shape, int dtype, const int *dims, const double *params);
/* Replace the taps (9/49/27 doubles) applied per sweep, bypassing the params mapping
 * (normalised-weights mode). */
int lora_plan_set_weights(lora_plan *plan, const double *weights, int count);
int lora_plan_get_weights(const lora_plan *plan, double *weights, int count);
int lora_plan_set_variant(lora_plan *plan, int variant);
/* Boundary condition applied by lora_plan_run (lora_boundary; lora_plan_step* stay raw sweeps). */
int lora_plan_set_boundary(lora_plan *plan, int boundary);
/* Boundary condition given to plans created afterwards on this thread, i.e. also to the host operators of group A
 * (what the CLIs' --bc flag sets).  Returns the previous value. */
int lora_set_default_boundary(int boundary);
/* Normalised-weights mode for plans created afterwards on this thread, i.e. also for the host operators of group A
 * (what the CLIs' --normalize flag sets): the operator's effective taps are divided by their sum, so that runs longer
 * than the fp64 range of the reference's integer taps allows stay finite (box2d3r overflows at step ~129, SURVEY B7;
 * BASELINE config 3 asks for 200).  Not reference behaviour.  Returns the previous value. */
int lora_set_default_normalize(int on);
/* Integer options.  Results never depend on them except where stated.
 *   steps_per_launch  0 auto / 1 / 2 (2D also 4 with the row-streaming kernel and 6 with the workgroup-row kernel, 3D fp64
 *                     also 3 with the plane-streaming kernel, 1D also 4, 8, 16, 32) : applications per launch in
 *                     lora_plan_run (temporal fusion).  2D auto: 6 (reference boundary; what six leave of a run is covered by
 *                     one launch of four and / or two), plain 49-tap tables included; 4 under the Dirichlet option.  1D auto:
 *                     the plan's own depth is 8 (lora_plan_stepk, slabs); lora_plan_run uses 16 from 32 sweeps on and
 *                     32 from 64 on
 *   rows_per_thread, panel_width, nt_store, fused_rows, persistent      2D tile shape / block->tile map / stores
 *   lowrank_valu      -1 auto / 0 off / 1 on / 2, 3 (plain / symmetric pyramid form) / 4 (rank-1 + correction instead of the
 *                     nested-profile form) : structured evaluation of the taps inside the fused 2D kernels (summation
 *                     order changes: identical while values are exact integers, ~1 ulp afterwards)
 *   z_chunk, fused_z_chunk             3D output planes per workgroup (single-sweep / fused kernels; 0 = auto)
 *   stream3           3D fp64 fused launches: 1 = plane-streaming kernel (kernels_3d_planes.hip: LDS-DMA plane ring, three
 *                     applications per launch for the 7-point star, two for the box), 0 = the two-application tile kernel,
 *                     -1 (default) = by grid size: three applications from ~1.2e8 points (star), the plane-streaming
 *                     kernel with two from ~2.4e7, the tile kernel below;
 *                     stream3_waves (0 automatic; 8 waves per workgroup, one per CU, or 4, two per CU), stream3_pipe (1 = one barrier per plane, two buffers
 *                     per published level; always on for two applications), stream3_async (1 = no workgroup barriers:
 *                     neighbour-wave counters in LDS; bit-identical, measured slower, off)
 *   separable         -1 auto / 0     exactly separable 3D taps as x/y/z passes: bf16 kernels (changes the fp32 summation
 *                     order; the oracle restates both orders, see lora_separable_3x3x3) and the fp64 plane-streaming
 *                     kernel (27 -> 9-10 multiply-adds per point; identical on integer data, ~1e-16 per sweep otherwise)
 *   cols_per_lane, lds_dma, fused_pipeline                              bf16 kernel variants
 *   graph             -1 auto / 0 / 1 : hipGraph replay of lora_plan_run
 *   scratch           -1 auto (= 1) / 0 / 1 : lora_plan_run may allocate one more grid (see there)
 *   mfma_split        bf16 matrix-pipe variant: 1 = hi + lo split of the intermediate (its contract), 0 = one bf16 rounding
 *   stream            2D fused launches: 1 = row-streaming kernel (wave-autonomous column strips, kernels_2d_stream.hip),
 *                     0 = tile kernel; stream_rows (output rows per chunk, 0 = auto), stream_depth (2..6 input rows in
 *                     flight per wave), stream_sync (one barrier per 7 rows keeps a workgroup's strips in step)
 *   wg                2D fused launches through the workgroup-row kernel (kernels_2d_wg.hip: 512-column rows owned by a
 *                     workgroup, the time levels pipelined over two groups of waves): -1 (default) = in plans that fuse six
 *                     applications per launch, where it also runs the four- / two-application tails; 0 never; 1 at every
 *                     depth (6 / 4 / 2).  wg_rows (output rows per chunk; 0 = one round of resident workgroups),
 *                     wg_edge_pct (how much shorter the chunks of the first / last column strip are, per cent; -1 = 60),
 *                     wg_prio (log2 of the time slice, in 10 ns ticks, of the alternating wave priorities that share a
 *                     CU evenly between its two workgroups; 0 = off)
 *   lanes3            3D fused launches through the register-resident kernels (kernels_3d_lanes.hip, kernels_3d_bf16_lanes.hip:
 *                     four applications per launch with the time levels in registers): -1 (default) by grid size (fp64 from
 *                     ~1e7 points, bf16 separable boxes from 1.2e7), 0 never, 1 always; steps_per_launch = 4 asks for them too
 *   spans3            3D register-resident kernels (fp64 and bf16, four applications per launch): how a launch is cut along
 *                     z.  0 = equal chunks per tile, 1 = spans (the line of all (tile, plane) pairs in equal pieces, one per
 *                     resident workgroup; csrc/spans.h), 2 = team spans (the line over tile rows, a piece per team of a
 *                     row's workgroups), -1 (default) = by measurement: spans on regions that fit the Infinity Cache, team
 *                     spans on bigger bf16 grids where they save steps, chunks otherwise.  Same bits every way
 *   torus             periodic boundary: 1 (default) = lora_plan_run goes through fused launches on a grid extended by a
 *                     ghost zone of periodic images on every side (two more buffers); 0 = single sweeps behind a halo wrap
 *                     each.  Grids smaller than a ghost zone and plans with steps_per_launch = 1 use the latter anyway
 * ("ablate", the load/store-removing timing experiment of round 1, exists only in -DLORA_DIAGNOSTICS builds of the
 * library; the shipped one answers LORA_EINVAL.)
 * lora_plan_get_option also reads the resolved "tapset", "variant", "fused_eval", "boundary". */
int lora_plan_set_option(lora_plan *plan, const char *key, int value);
int lora_plan_get_option(const lora_plan *plan, const char *key, int *value);
size_t lora_plan_padded_bytes(const lora_plan *plan);
/* Name of the kernel a sweep of this plan launches (for matching rocprof rows). */
const char *lora_plan_kernel_name(const lora_plan *plan);
/* The same plus every resolved option that selects the kernel instantiation or its launch geometry, e.g.
 * "stencil2d_wg_kernel[eval=7,k=6,rows=0,edge=-1,prio=12,bc=0]": the key measured per-launch HBM traffic is filed under
 * (profiles/pmc_traffic.json), so that a changed option can never be paired with a stale measurement. */
const char *lora_plan_kernel_signature(const lora_plan *plan);

/* One kernel application: interior of d_out <- stencil(d_in).  Halo cells of d_out are not
 * touched (2d/gpu.cu:266-271). */
int lora_plan_step(lora_plan *plan, const void *d_in, void *d_out, void *stream);
/* The same restricted to outermost-dimension interior indices [begin, end) (rows in 2D, planes
 * in 3D, points in 1D) -- used to compute slab boundaries first and overlap the halo exchange
 * with the rest.  begin must be a multiple of lora_plan_region_granularity() (2 in 1D, 1 otherwise). */
int lora_plan_step_region(lora_plan *plan, const void *d_in, void *d_out, int begin, int end, void *stream);
int lora_plan_region_granularity(const lora_plan *plan);
/* TWO kernel applications in one launch (temporal fusion; tiled 2D direct-variant and 3D plans): the intermediate time level
 * lives in LDS and its halo cells are taken as 0 -- the state of the reference driver's second buffer --
 * so d_in must be an even time level (halo = the caller's input halo).  Interior of d_out <- stencil(stencil(d_in)).
 * lora_plan_run uses it when the plan's resolved "steps_per_launch" is 2 (3D tile / bf16 kernels, 2D under the Dirichlet
 * option). */
int lora_plan_step2(lora_plan *plan, const void *d_in, void *d_out, void *stream);
int lora_plan_step2_region(lora_plan *plan, const void *d_in, void *d_out, int begin, int end, void *stream);
/* The plan's resolved "steps_per_launch" applications in ONE launch: 1 = lora_plan_step, 2 = lora_plan_step2 (2D / 3D),
 * 4 / 6 in 2D, 3 in 3D fp64, 2 / 4 / 8 in 1D (intermediate levels in LDS; halo cells of odd intermediate levels are 0, of even ones the source
 * buffer's halo -- the state the step-by-step driver would leave).  d_in must be an even time level.
 * lora_plan_run uses an even number of these and finishes with single sweeps; three-application 3D launches have an
 * odd count and run on the reference's alternating buffer state instead (launch k reads buffer k mod 2, whose halo
 * is the caller's for even k and zero for odd k), any number of them. */
int lora_plan_stepk(lora_plan *plan, const void *d_in, void *d_out, void *stream);
int lora_plan_stepk_region(lora_plan *plan, const void *d_in, void *d_out, int begin, int end, void *stream);
/* `napps` applications in one launch: 1, the plan's own depth, or one of the shallower depths its runs use for their tails
 * (2D workgroup-row kernel: 4 and 2 under a six-application plan; 1D: powers of two below the plan's depth; else 2).
 * LORA_EUNSUPPORTED for a depth the plan's kernels do not have. */
int lora_plan_stepn_region(lora_plan *plan, int napps, const void *d_in, void *d_out, int begin, int end, void *stream);
/* The same over TWO disjoint ranges [begin0, end0) and [begin1, end1): what a slab or block driver sweeps behind its
 * deferred wait (the two ends of its share).  One launch where the kernel family takes two ranges (the register-resident 3D
 * kernels), two launches otherwise; an empty range is skipped. */
int lora_plan_stepn_region2(lora_plan *plan, int napps, const void *d_in, void *d_out, int begin0, int end0, int begin1, int end1,
                            void *stream);
/* Test support, no device needed: replays on the host the decode the register-resident 3D kernels run for a launch cut into
 * spans (team = 0), team spans (team = 1) or team spans with the rim columns cut finer (team = 2; csrc/spans.h) over tiles_x x tiles_y tiles and `depth` planes on `slots`
 * resident workgroups with S steps of start per segment.  cover[(ty * tiles_x + tx) * depth + z] is incremented once per
 * segment that sweeps the pair (the caller zeroes it): every entry must come out 1. */
int lora_debug_span_cover(int tiles_x, int tiles_y, int depth, int S, int slots, int team, int *cover, int *workgroups, int *max_steps);
/* Halo cells of a padded device array (every cell outside the interior, any shape / dtype of the plan): copied from
 * d_src (LORA_HALO_COPY), zeroed (LORA_HALO_ZERO) or wrapped from d_dst's own opposite interior edges
 * (LORA_HALO_WRAP, periodic; LORA_EUNSUPPORTED if an extent is smaller than its halo).  What lora_plan_run uses for
 * the boundary options and the fused launches' halo bookkeeping; exported for drivers that step themselves. */
enum lora_halo_mode { LORA_HALO_COPY = 0, LORA_HALO_ZERO = 1, LORA_HALO_WRAP = 2 };
int lora_plan_halo(lora_plan *plan, void *d_dst, const void *d_src, int mode, void *stream);
/* rows x cols fp64 elements from one strided device array to another (leading dimensions in elements): the pack / unpack
 * kernel of a 2-D block decomposition's COLUMN ghost zones (lorastencil_amd/blocks.py) -- a slab's ghost rows
 * are contiguous and need none.  Asynchronous on `stream`. */
int lora_copy_block_f64(void *d_dst, long dst_ld, const void *d_src, long src_ld, long rows, long cols, void *stream);
/* The time-step driver (2d/gpu.cu:544-546): `times` applications ping-ponging between the two
 * buffers starting from d_buf0; the result is in buffer [times % 2] (the other buffer's interior is
 * unspecified).  The caller must have put the padded input in d_buf0 and zeros in d_buf1 to get the
 * reference semantics.  Fused launches move K time levels per buffer flip, so an ODD number of them
 * would leave the data in the wrong buffer: the plan then routes the last two through a scratch grid of
 * its own (one more padded array, allocated on first need, freed with the plan; option "scratch" = 0
 * trades it for a shorter launch schedule instead). */
int lora_plan_run(lora_plan *plan, void *d_buf0, void *d_buf1, int times, void *stream);
/* Allocates now what lora_plan_run(plan, ..., times, ...) would allocate on first need (the scratch grid of a schedule with
 * an odd number of fused launches, the extended grid of a periodic run), so that a timed or latency-sensitive first run
 * does not pay for it.  Optional and idempotent. */
int lora_plan_prepare_run(lora_plan *plan, int times);
/* lora_plan_run with HIP events recorded on `stream` around its two kinds of launches -- the fused multi-application
 * launches and the single-sweep tail -- so that a caller can quote the average duration of the dominant kernel over
 * the very region it timed (bench.py's roofline).  Launches directly (no hipGraph) and BLOCKS until the run has
 * finished.  Halo bookkeeping kernels (O(surface)) are inside the fused segment. */
typedef struct lora_run_profile {
    int fused_launches;        /* launches of the K-application kernel                               */
    int apps_per_fused_launch; /* K (1 if the plan does not fuse)                                    */
    int two_launches;          /* shallower fused launches of the tail: 2D: four and / or two applications; 1D: K/2 .. 2 */
    int single_launches;       /* single-sweep launches (the whole run if K = 1)                     */
    float fused_ms;            /* event time of the K-application launches (+ the halo copy)         */
    float two_ms;              /* ... of the shallower fused launches (+ the halo reset)             */
    float single_ms;           /* ... of the single-sweep segment                                    */
} lora_run_profile;
int lora_plan_run_profiled(lora_plan *plan, void *d_buf0, void *d_buf1, int times, void *stream,
                           lora_run_profile *profile);
void lora_plan_destroy(lora_plan *plan);

/* ========================================================================================
 * D. Multi-GPU slabs (NEW: the reference is single-GPU).  The grid is cut along its outermost interior
 *    dimension into one slab per GPU; the time-step loop (2d/gpu.cu:544-546) runs on every slab with a nearest-
 *    neighbour ghost-zone exchange over RCCL (ncclSend / ncclRecv in one group, on a communication stream) overlapped
 *    with the interior sweeps.  N slabs == 1 GPU bit for bit (same kernels, same per-point arithmetic).
 *    Usual form: one process per GPU, each creating ONE slab with its rank and an RCCL communicator; the CLIs' --gpus N
 *    drive N slabs of one process (lora_run_host_multi).
 * ====================================================================================== */
typedef struct lora_slab lora_slab;

/* How neighbouring slabs exchange ghost rows.  All calls are made from the driving host thread; send / recv enqueue on
 * `stream` (a hipStream_t) and must not block the host; every exchange of a launch sits between one group_begin and
 * one group_end (called on the first slab's table).  `peer` is a rank of the decomposition. */
typedef struct lora_slab_comm {
    void *ctx;
    int (*group_begin)(void *ctx);
    int (*send)(void *ctx, const void *d_buf, size_t bytes, int peer, void *stream);
    int (*recv)(void *ctx, void *d_buf, size_t bytes, int peer, void *stream);
    int (*group_end)(void *ctx);
} lora_slab_comm;
/* RCCL: ctx = an ncclComm_t of this rank (ncclCommInitRank / ncclCommInitAll, or torch's); librccl.so is loaded on
 * first use, the engine library does not link it. */
int lora_slab_comm_rccl(lora_slab_comm *out, void *nccl_comm);
/* In-process loopback for `nranks` slabs that live in ONE process on ONE device: messages become device-to-device
 * copies ordered by events.  Fills out[0 .. nranks-1].  For tests / rehearsals on a one-GPU box. */
int lora_slab_comm_loopback(lora_slab_comm *out, int nranks);

enum lora_slab_flags {
    LORA_SLAB_NO_OVERLAP = 1,  /* sweep the whole slab, then exchange (no boundary-first split); the default
                                  in 2D and 3D: a fused launch is ONE round of workgroups sized to its region,
                                  so a thin boundary strip costs a third of a whole-slab launch (measured:
                                  DESIGN section 6)                                                           */
    LORA_SLAB_NO_DEFER = 2,    /* wait for an exchange at the end of its launch instead of inside the next   */
    LORA_SLAB_NO_FUSION = 4,   /* single sweeps only                                                         */
    LORA_SLAB_OVERLAP = 16,    /* boundary strips first, exchange overlapped with the interior (the default in
                                  1D; forces it in 2D / 3D)                                                   */
    LORA_SLAB_RING_OF_ONE = 8  /* nranks == 1: the slab is its own neighbour (periodic along the split
                                  dimension) -- the complete exchange path on one GPU; with the reference boundary
                                  a rehearsal of one rank's share of an N-GPU run, not a physical result      */
};
typedef struct lora_slab_desc {
    int shape, dtype;          /* lora_shape, lora_dtype                                                      */
    int global_dims[3];        /* interior extents of the WHOLE grid, outermost first                         */
    const double *params;      /* as lora_plan_create (NULL: the reference harness's table)                   */
    const double *weights;     /* nullable: taps applied per sweep, bypassing the params mapping              */
    int rank, nranks;          /* this slab and the number of slabs (GPUs)                                    */
    int device;                /* HIP device the slab lives on                                                */
    int exchange_every;        /* launches between ghost-zone refreshes; 0 = auto (by slab thickness) */
    int boundary;              /* LORA_BC_REFERENCE or LORA_BC_DIRICHLET                                      */
    int flags;                 /* lora_slab_flags                                                             */
    const char *options;       /* nullable "key=value,key=value": plan options, applied before the ghost depth is fixed */
} lora_slab_desc;
typedef struct lora_slab_info_t {
    int begin, end;            /* global interior range of the slab's own rows                                */
    int ghost, ghost_top, ghost_bottom;
    int apps_per_launch, exchange_every, steps_done;
    int local_dims[3];         /* interior extents of the local array (own + ghost rows)                      */
    long launches, exchanges;
    size_t local_bytes;        /* bytes of one local padded buffer                                            */
} lora_slab_info_t;

/* `comm` may be NULL for nranks == 1 without the ring flag.  Allocates the slab's two local buffers and streams. */
int lora_slab_create(lora_slab **slab, const lora_slab_desc *desc, const lora_slab_comm *comm);
void lora_slab_destroy(lora_slab *slab);
int lora_slab_info(const lora_slab *slab, lora_slab_info_t *info);
/* buffer 0 <- this slab's rows of the padded GLOBAL host array (ghost rows and pads included), buffer 1 <- 0 */
int lora_slab_load(lora_slab *slab, const void *host_global_padded);
/* the same from a LOCAL padded device array (own + ghost rows + pads), e.g. data generated on the device;
 * call lora_slab_refresh_ghosts afterwards */
int lora_slab_load_device(lora_slab *slab, const void *d_local_padded);
int lora_slab_refresh_ghosts(lora_slab *slab);
/* `times` kernel applications (fused launches from even time levels, single sweeps otherwise); asynchronous */
int lora_slab_run(lora_slab *slab, int times);
int lora_slab_sync(lora_slab *slab);
/* own rows (+ the pad rows at a global edge) of the current time level -> their place in the padded GLOBAL host array */
int lora_slab_store(lora_slab *slab, void *host_global_padded);
/* which = 0 / 1: the two local device buffers; anything else: the one holding the current time level */
void *lora_slab_buffer(lora_slab *slab, int which);
void *lora_slab_stream(lora_slab *slab); /* the hipStream_t its sweeps run on */
lora_plan *lora_slab_plan(lora_slab *slab);
/* Several slabs of ONE decomposition driven by this host thread (one process, N devices): launches and exchanges are
 * interleaved across the slabs, every exchange inside one group. */
int lora_slab_run_many(lora_slab **slabs, int n, int times);
int lora_slab_refresh_ghosts_many(lora_slab **slabs, int n);
/* Group A's generic operator on `ngpus` devices of this node: one slab per device, RCCL from ncclCommInitAll.
 * (LORA_SLAB_LOOPBACK=1 in the environment: all slabs on device 0 with the loopback exchange -- one-GPU rehearsal.) */
int lora_run_host_multi(int shape, int dtype, const void *in, void *out, const double *params, int times,
                        const int *dims, int ngpus, int quiet, lora_run_info *info);

/* ========================================================================================
 * E. Two-axis block decomposition (NEW).  A Pa x Pb process grid cuts the two OUTER dimensions of the grid --
 *    rows x columns of the 2D shapes, planes x rows (z x y) of the 3D ones -- into one block per GPU.  A block's local
 *    array carries ghost zones of radius x applications-per-launch x E cells on every cut side; they are refreshed every E
 *    launches in two phases (axis B: strips packed by the block-copy kernel; then axis A: whole rows / planes in place,
 *    B-side ghost cells included, so corners need no diagonal message) over the same callback table as the slabs
 *    (peer = ia * Pb + ib), the second phase hidden behind the next launch's interior (deferred wait).  The loop that is
 *    distributed is the reference's time-step loop (2d/gpu.cu:544-546, 3d/gpu_star.cu:177-181); the reference itself has
 *    no multi-GPU path.  Reference boundary; fp64, and bf16 for the 3D shapes.  N blocks == 1 GPU bit for bit.
 * ====================================================================================== */
typedef struct lora_block lora_block;
typedef struct lora_block_desc {
    int shape, dtype;          /* a 2D or 3D lora_shape; lora_dtype                                          */
    int global_dims[3];        /* interior extents of the WHOLE grid, outermost first                         */
    int grid[2];               /* Pa x Pb: parts of the outermost (axis A) and of the second dimension (B)    */
    int coords[2];             /* this block: (ia, ib); its rank in the callback table is ia * Pb + ib        */
    const double *params;      /* as lora_plan_create (NULL: the reference harness's table)                   */
    const double *weights;     /* nullable: taps applied per sweep                                            */
    int device;                /* HIP device the block lives on                                               */
    int exchange_every;        /* launches between ghost-zone refreshes; 0 = 2; clipped to the thinnest block */
    int flags;                 /* LORA_SLAB_NO_DEFER, LORA_SLAB_NO_FUSION                                     */
    const char *options;       /* nullable "key=value,key=value": plan options                                */
} lora_block_desc;
typedef struct lora_block_info_t {
    int own_begin[2], own_end[2];    /* global interior ranges of the block's own cells along axis A / B     */
    int ghost, ghost_lo[2], ghost_hi[2];
    int apps_per_launch, exchange_every, steps_done;
    int local_dims[3];               /* interior extents of the local array (own + ghost cells)              */
    long launches, exchanges;
    size_t local_bytes;              /* bytes of one local padded buffer                                     */
    size_t bytes_per_refresh;        /* bytes this block sends per ghost-zone refresh (both phases)          */
} lora_block_info_t;
/* `comm` may be NULL for a 1 x 1 grid.  Allocates the block's two local buffers, its pack strips and streams. */
int lora_block_create(lora_block **block, const lora_block_desc *desc, const lora_slab_comm *comm);
void lora_block_destroy(lora_block *block);
int lora_block_info(const lora_block *block, lora_block_info_t *info);
/* buffer 0 <- this block's cells of the padded GLOBAL host array (ghost cells and pads included), buffer 1 <- 0 */
int lora_block_load(lora_block *block, const void *host_global_padded);
/* `times` kernel applications (fused launches from even time levels, single sweeps otherwise); asynchronous */
int lora_block_run(lora_block *block, int times);
/* several blocks of ONE decomposition driven by this host thread: every exchange phase inside one group */
int lora_block_run_many(lora_block **blocks, int n, int times);
int lora_block_sync(lora_block *block);
/* own cells (+ the pads at a global edge) of the current time level -> their place in the padded GLOBAL host array */
int lora_block_store(lora_block *block, void *host_global_padded);
void *lora_block_buffer(lora_block *block, int which);
void *lora_block_stream(lora_block *block);
lora_plan *lora_block_plan(lora_block *block);
/* Group A's generic operator on a grid[0] x grid[1] grid of blocks, one per device of this node (RCCL from
 * ncclCommInitAll; LORA_SLAB_LOOPBACK=1: all blocks on device 0 with the loopback exchange).  What the CLIs' --grid=AxB runs. */
int lora_run_host_blocks(int shape, int dtype, const void *in, void *out, const double *params, int times,
                         const int *dims, const int *grid, int quiet, lora_run_info *info);

/* ========================================================================================
 * C. Host helpers on the path.
 * ====================================================================================== */
/* The params tables the reference harness builds (1d/main.cu:77-78, 2d/main.cu:139-195,
 * 3d/main.cu:112-125).  Returns the tap count. */
int lora_default_params(int shape, double *params);
/* The taps the reference operator applies for `params` (see group A).  Returns the count. */
int lora_effective_weights(int shape, const double *params, double *weights);
/* Low-rank factor precompute of the box2d operator (2d/gpu.cu:280-350): pyramid peeling of a
 * symmetric 7x7 matrix into rank-1 terms u[t] (x) v[t], t = 0..3 (4x7 doubles each, row-major).
 * `residual_max` (nullable) receives max |params - sum_{t<3} u_t v_t^T|: the part the
 * reference silently drops. */
int lora_factorize_7x7(const double *params, double *u, double *v, double *residual_max);

/* Rank-revealing factorisation of ANY 7x7 tap matrix (one-sided Jacobi SVD):
 * weights = sum_k u[k] (x) v[k], u[k] = sigma_k a_k (vertical profile), v[k] = b_k (horizontal profile), k = 0..6 in
 * order of decreasing singular value sigma[k] (7x7, 7x7 and 7 doubles).  Keeping `r` terms leaves a residual of
 * spectral norm sigma[r].  The MFMA variant uses it for taps the pyramid factoriser cannot take. */
int lora_svd_7x7(const double *weights, double *u, double *v, double *sigma);

/* Exact rank-1 test of 3x3x3 taps as the bf16 sweeps see them (cast to fp32): returns 1 and fills
 * cba9 = {c[0..2] (x), b[0..2] (y), a[0..2] (z)} when fl(fl(a[dz] * b[dy]) * c[dx]) == w[dz][dy][dx] for all 27
 * taps, else 0.  Every box3d1r the reference can express passes (gpu_box_3d1r honours params[0..2] only,
 * 3d/gpu_box.cu:143-226); bf16 plans then evaluate the sweep as x-, y- and z-passes (plan option "separable",
 * default on; the evaluation order is documented in kernels_3d_bf16.hip and restated by the oracle). */
int lora_separable_3x3x3(const double *weights27, float *cba9);

/* bf16 <-> double on the host (round-to-nearest-even; bf16 values are bit patterns in uint16_t). */
void lora_f64_to_bf16(const double *src, uint16_t *dst, size_t count);
void lora_bf16_to_f64(const uint16_t *src, double *dst, size_t count);

/* glibc rand() stream (TYPE_3, seed 1 = the reference's un-seeded rand()) so that inputs are
 * identical on any libc.  Fill = (double)(rand() % mod): 1d/main.cu:105-109 (mod 10000),
 * 2d/main.cu:232-236 and 3d/main.cu:164-168 (mod 100). */
typedef struct lora_rng {
    int32_t r[34];
    int32_t pos;
} lora_rng;
void lora_rng_seed(lora_rng *g, unsigned seed);
int lora_rng_next(lora_rng *g);
void lora_fill_rand(double *dst, size_t count, int mod, lora_rng *g);

#ifdef __cplusplus
}
#endif
#endif /* LORASTENCIL_H */
