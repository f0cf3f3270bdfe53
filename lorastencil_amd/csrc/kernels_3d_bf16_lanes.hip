// kernels_3d_bf16_lanes.hip -- FOUR applications of an exactly separable 3D radius-1 box per launch on bf16 grids, the
// time levels held in registers (BASELINE config 5: box3d1r 768^3 bf16; no reference counterpart -- the reference is
// fp64-only; the loop it replaces is 3d/gpu_box.cu:105-140 inside the time-step loop :208-212).
//
// Where round 3 left bf16 (profiles/r02_box3d1r_bf16_pmc.json): stencil3d_bf16_fused2_kernel applies two
// sweeps per launch with level 1 as a bf16 LDS tile.  It is bound by the fp32 vector pipe -- 15.3 lane-instructions per
// point and level where the arithmetic of the separable form is 9 multiply-adds and 1.5 conversions -- because every level
// re-reads 12-element LDS windows, converts each element for up to three column pairs and rebuilds odd-aligned pairs.
//
// What the vector pipe charges (tools/probes/fp32_rate_probe.hip, profiles/r04_fp32_rate_probe.txt; SIMD time per wave
// instruction at 1 / 2 / 4 waves per SIMD): v_fma_f32 / v_fmac_f32 5.0 / 2.5 / 1.3, v_pk_fma_f32 5.8 / 4.8 / 3.4 (two points:
// no cheaper per point than the plain form, and only when four waves share the SIMD does the plain form reach its rate --
// ONE wave issues an instruction every ~5 cycles whatever it is), DPP forms, shifts, v_cvt_pk_bf16_f32 and v_fma_f64 3.3.
// So the kernel wants FOUR waves per SIMD (<= 128 VGPRs), plain fp32 multiply-adds, and few of the slow forms.
//
// As in kernels_3d_lanes.hip (fp64), a level never goes to LDS as a tile:
//   * A lane owns the same R = 4 rows x 2 columns of a 64 x 128 tile at EVERY level: one packed bf16 pair (a dword) per
//     row; 16 waves (1024 threads) share the tile, one workgroup per CU.  x-neighbours are the neighbouring lanes'
//     registers (DPP wave_shr:1 / wave_shl:1 folded into the multiply-add that consumes them), y-neighbours the lane's own
//     rows and, across waves, the x-PASSED form of each wave's first / last row (8 bytes per lane and level through LDS);
//     z is streamed: per level and point TWO fp32 partial sums live between steps -- in a step the older one takes its
//     dz = 2 tap, is rounded to bf16 (the level's completed plane) and is re-opened in place with the dz = 0 tap of the
//     same U; the other takes its dz = 1 tap.  The two slots swap roles every step (the loop is unrolled by two).
//   * The levels are SKEWED by one step: level l + 1 consumes in step p the plane level l completed in step p - 1 (a
//     packed bf16 plane: R registers per level).  All K levels of a step are then independent of each other: every level
//     publishes its edge rows before ONE barrier per step, and the scheduler has K independent chains to interleave.
//     Price: K - 1 more steps per z-chunk (zc + 3 K - 1 instead of zc + 2 K).
//   * Numerics are the contract of kernels_3d_bf16.hip / oracle_step_3d_bf16_sep: T = fma(c2, x+, fma(c1, x0, c0 x-))
//     along x, U likewise along y over T, out along z over U, all fp32, ONE round-to-nearest-even to bf16 per level
//     (v_cvt_pk_bf16_f32) -- so a launch is bit-identical to K single sweeps.  Unpacking a level's plane for the next level
//     is a shift and a mask per pair.
//   * Input planes: every wave fetches its own four rows of the next plane with ONE 1 KiB LDS-DMA (16 bytes per lane,
//     whole 256-byte row pieces) into a private ring (two slots; five for K = 4, whose EDGE steps look three planes back)
//     and picks its dwords up from there -- no staging registers.
//     Output planes leave as dword stores through a range-checked buffer descriptor per row, issued at the START of the
//     next step right behind the step's one s_waitcnt vmcnt(0): everything that wait covers (the DMA of this step's plane,
//     the stores of the plane before) was issued a whole step earlier, and no count relies on the order in which loads
//     and stores complete.
//   * Halo semantics: cells of an intermediate level outside the interior are 0 at odd levels and the source
//     buffer's own halo value at even ones (level 2 of four).  Rim tiles, and every tile in the steps whose planes lie
//     outside the z range, run an EDGE copy of the step that forces them with a bit-field insert per pair; the level-2
//     values are the input's own values three planes back (while fused launches run every buffer carries buffer 0's halo):
//     the lane's dwords of a plane that is still in the input ring.
//   * A segment's first 3 (K - 1) and last K - 1 steps run a FILL copy of the step that skips the levels with no work yet /
//     any more; the launch is cut along z into chunks, spans or team spans (spans.h).
//
// Taps: exactly separable 27-point boxes (TAPS3D_SEP: every box3d1r the reference's API can express, 3d/gpu_box.cu:158-164).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "device_common.h"
#include "spans.h"

// Timing experiments for tools/probes/bf16_lanes_ablate.hip (results are WRONG with bits 1 .. 16 set): 1 no barriers, 2 no
// stores, 4 no plane loads after the first, 8 no EDGE steps, 16 only EDGE steps, 256 no FILL copy of the first / last steps, 512 team spans with equal pieces in every tile column
// (right results); 32 = the taps as scalar operands (right
// results, the form the first version of this kernel had: an fp32 instruction with a scalar operand issues at half rate)
#ifndef LORA_BL_ABLATE
#define LORA_BL_ABLATE 0
#endif
// -DLORA_BL_STAMP=n (tools/probes/bf16_lanes_ablate.hip): every wave adds up the s_memtime cycles phase n of its steps takes
// (1 upper levels ahead of the barrier, 2 barrier, 3 behind it, 4 wait + stores + DMA, 5 lower levels ahead, 6 barrier,
// 7 behind) and leaves the sum and its step count in a debug buffer; the shipped library has none of it
#ifndef LORA_BL_STAMP
#define LORA_BL_STAMP 0
#endif
// bit 128 of LORA_BL_ABLATE: the y- and z-passes of a lane's column pair as packed instructions (v_pk_mul_f32 /
// v_pk_fma_f32: two points each, 29 % fewer vector instructions per step; same results).  Measured SLOWER in this kernel:
// 1034 - 1039 us per launch at 768^3 against 865 - 887 for the plain form (profiles/r04_bf16_lanes_packed_ab.txt), although
// the bare instruction sequence is 13 % faster packed (profiles/r04_bf16_level_probe.txt) -- kept for the record only.
#define LORA_BL_PACKED ((LORA_BL_ABLATE & 128) != 0)
#if LORA_BL_ABLATE & 32
#define LORA_BL_TAP "s"
#else
#define LORA_BL_TAP "v"
#endif

namespace lora {

namespace {

constexpr int kTileW = 128;  // columns of a tile: 64 lanes x 2
constexpr int kOutW = 120;   // output columns of a tile: lanes 2 .. 61

typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct ArgsBL {
    const u16 *in;
    u16 *out;
    int h, m, n;
    int ld;
    long plane;
    int z_begin, z_end;
    int z_begin2, z_end2;  // a second range of planes in the same launch (chunks only): the two end regions of a slab
    int zc;
    int tiles_x, tiles_y;
    Spans sp;  // zc == 0: spans (spans.h)
    int team;  // zc == 0: 0 = a span per workgroup over all tiles; TX = a span per TEAM of the TX workgroups of a tile row
    // team spans: the first / last tile COLUMN (rim tiles: EDGE steps, 5 % longer each) is cut into team_nr pieces (Spans sp2),
    // the columns between into team_ni (Spans sp); team_nr == team_ni: whole teams in lockstep
    int team_ni, team_nr;
    Spans sp2;
#if LORA_BL_STAMP
    long long *stamps;  // [workgroup][wave][2]: cycles of the stamped phase, steps
#endif
#ifdef LORA_BL_TIMELINE
    long long *timeline;  // probe builds: per workgroup {first start, last end} on the 100 MHz clock, steps run, segments | rim << 32
#endif
};
struct TapsSep {
    float c[3], b[3], a[3];
};

// The x-pass of a lane's four rows (four packed bf16 pairs -> eight fp32 sums), written out as the instructions it is.
// Per row:  x0, x1 = the pair's halves as fp32 (a shift, a mask)
//   t0 = fma(c2, x1, fma(c1, x0, c0 * x1[lane - 1]))        t1 = fma(c2, x0[lane + 1], fma(c1, x1, c0 * x0))
// with the neighbour lanes' values taken as DPP operands of the multiply / multiply-add that consumes them (wave_shr:1 /
// wave_shl:1, 0 beyond the wave's ends: lanes 0 and 63 hold cells that are never valid beyond level 0).  Why by hand:
// hipcc folds a DPP move into v_mul_f32 but not into v_fmac_f32, its SLP pass packs the pair's multiply-adds into
// v_pk_fma_f32 behind register moves -- both cost more than they save here (tools/probes/fp32_rate_probe.hip) -- and it
// pads every asm statement with an s_nop, so the four rows are ONE statement.  A DPP operand must not be read within two
// instructions of the VALU instruction that wrote it (the hardware does not interlock that, and the compiler cannot see into
// an asm statement): the order below keeps two or more between them.  c0 / c2 are passed in VECTOR registers (a DPP
// instruction takes no scalar operand), c1 in a scalar one.
#define LORA_X1(N) "v_lshlrev_b32_e32 %[x0" #N "], 16, %[in" #N "]\n\t"
#define LORA_X2(N) "v_and_b32_e32 %[x1" #N "], 0xffff0000, %[in" #N "]\n\t"
#define LORA_X3(N) "v_mul_f32_e32 %[t1" #N "], %[c0], %[x0" #N "]\n\t"
#define LORA_X4(N) "v_fmac_f32_e32 %[t1" #N "], %[c1], %[x1" #N "]\n\t"
#define LORA_X5(N) "v_mul_f32_dpp %[t0" #N "], %[x1" #N "], %[c0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define LORA_X6(N) "v_fmac_f32_e32 %[t0" #N "], %[c1], %[x0" #N "]\n\t"
#define LORA_X7(N) "v_fmac_f32_e32 %[t0" #N "], %[c2], %[x1" #N "]\n\t"
#define LORA_X8(N) "v_fmac_f32_dpp %[t1" #N "], %[x0" #N "], %[c2] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#if LORA_BL_ABLATE & 64  // (A/B: row after row -- every multiply-add right behind the instruction it depends on)
#define LORA_XROW(N) LORA_X1(N) LORA_X2(N) LORA_X3(N) LORA_X4(N) LORA_X5(N) LORA_X6(N) LORA_X7(N) LORA_X8(N)
#define LORA_XROWS LORA_XROW(a) LORA_XROW(b) LORA_XROW(c) LORA_XROW(d)
#else  // the four rows interleaved: four independent chains, a dependent instruction four slots behind its producer
#define LORA_XALL(X) X(a) X(b) X(c) X(d)
#define LORA_XROWS LORA_XALL(LORA_X1) LORA_XALL(LORA_X2) LORA_XALL(LORA_X3) LORA_XALL(LORA_X4) LORA_XALL(LORA_X5) LORA_XALL(LORA_X6) LORA_XALL(LORA_X7) LORA_XALL(LORA_X8)
#endif
__device__ __forceinline__ void xpass_rows(const unsigned (&in)[4], float c0v, float c1, float c2v, f2 (&t)[4]) {
    float x0a, x1a, x0b, x1b, x0c, x1c, x0d, x1d;
    asm(LORA_XROWS
        : [x0a] "=&v"(x0a), [x1a] "=&v"(x1a), [t0a] "=&v"(t[0].x), [t1a] "=&v"(t[0].y),
          [x0b] "=&v"(x0b), [x1b] "=&v"(x1b), [t0b] "=&v"(t[1].x), [t1b] "=&v"(t[1].y),
          [x0c] "=&v"(x0c), [x1c] "=&v"(x1c), [t0c] "=&v"(t[2].x), [t1c] "=&v"(t[2].y),
          [x0d] "=&v"(x0d), [x1d] "=&v"(x1d), [t0d] "=&v"(t[3].x), [t1d] "=&v"(t[3].y)
        : [ina] "v"(in[0]), [inb] "v"(in[1]), [inc] "v"(in[2]), [ind] "v"(in[3]), [c0] "v"(c0v), [c1] LORA_BL_TAP(c1), [c2] "v"(c2v));
}
// The taps in vector registers.  Plain form: nine.  Packed form: the x-pass's three and the y / z taps as three register
// PAIRS (b0, b1), (b2, a0), (a1, a2) -- a packed instruction takes its tap from either half of a pair (op_sel), so six taps
// cost six registers, not the twelve of six splat pairs.
struct TapsV {
    float c0, c1, c2, b0, b1, b2, a0, a1, a2;
    f2 b01, b2a0, a12;
};
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned round_pair(const f2 o) {  // one v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(o, bf2));
}
// y-pass and z-pass of TWO rows of a lane (two points each), per point in the contract's order:
//   u = fma(b2, t[r + 1], fma(b1, t[r], b0 * t[r - 1]))
//   o = fma(a2, u, cur)   the level's completed value (cur had its dz = 0, 1 taps);   nxt = fma(a1, u, nxt);   cur = a0 * u
// and the completed values leave as packed bf16 pairs.
// Packed form: a lane's two columns sit in an aligned register pair, every line above is ONE v_pk instruction on the pair
// (two points), the tap is the low or high half of a tap pair (op_sel / op_sel_hi on the first operand).  By hand because
// hipcc materialises every splat tap as a register pair of its own (twelve registers more: the kernel spills at its 128).
// The two rows are interleaved: a packed instruction is never read by the very next one (the pipe forwards a packed result
// to a dependent packed instruction only with an instruction between them -- hipcc itself puts an s_nop there).
// Plain form: the same arithmetic as v_mul / v_fmac / v_fma_f32 per point; the completed value as a THREE-address v_fma_f32
// (with the two-address v_fmac_f32 the compiler ties it to cur's register and then needs a copy per point and step to
// bring the re-opened sum back around the loop).
#define LORA_LO "op_sel_hi:[0,1,1]"                  /* first operand: its low half for both points */
#define LORA_HI "op_sel:[1,0,0] op_sel_hi:[1,1,1]"   /* first operand: its high half for both points */
__device__ __forceinline__ void yz_rows2(const f2 tA_m, const f2 tA_0, const f2 tA_p, const f2 tB_m, const f2 tB_0, const f2 tB_p, const TapsV &T,
                                         f2 &curA, f2 &nxtA, f2 &curB, f2 &nxtB, unsigned &vA, unsigned &vB) {
#if LORA_BL_PACKED
    f2 uA, uB, oA, oB;
    asm("v_pk_mul_f32 %[uA], %[b01], %[tAm] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[uB], %[b01], %[tBm] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[uA], %[b01], %[tA0], %[uA] " LORA_HI "\n\t"
        "v_pk_fma_f32 %[uB], %[b01], %[tB0], %[uB] " LORA_HI "\n\t"
        "v_pk_fma_f32 %[uA], %[b2a0], %[tAp], %[uA] " LORA_LO "\n\t"
        "v_pk_fma_f32 %[uB], %[b2a0], %[tBp], %[uB] " LORA_LO "\n\t"
        "v_pk_fma_f32 %[oA], %[a12], %[uA], %[curA] " LORA_HI "\n\t"
        "v_pk_fma_f32 %[oB], %[a12], %[uB], %[curB] " LORA_HI "\n\t"
        "v_pk_fma_f32 %[nxtA], %[a12], %[uA], %[nxtA] " LORA_LO "\n\t"
        "v_pk_fma_f32 %[nxtB], %[a12], %[uB], %[nxtB] " LORA_LO "\n\t"
        "v_pk_mul_f32 %[curA], %[b2a0], %[uA] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %[curB], %[b2a0], %[uB] op_sel:[1,0] op_sel_hi:[1,1]"
        : [uA] "=&v"(uA), [uB] "=&v"(uB), [oA] "=&v"(oA), [oB] "=&v"(oB), [curA] "+v"(curA), [nxtA] "+v"(nxtA), [curB] "+v"(curB), [nxtB] "+v"(nxtB)
        : [tAm] "v"(tA_m), [tA0] "v"(tA_0), [tAp] "v"(tA_p), [tBm] "v"(tB_m), [tB0] "v"(tB_0), [tBp] "v"(tB_p), [b01] "v"(T.b01), [b2a0] "v"(T.b2a0),
          [a12] "v"(T.a12));
    vA = round_pair(oA);
    vB = round_pair(oB);
#else
    const f2 t3[2][3] = {{tA_m, tA_0, tA_p}, {tB_m, tB_0, tB_p}};
    f2 *const cur[2] = {&curA, &curB}, *const nxt[2] = {&nxtA, &nxtB};
    unsigned *const v[2] = {&vA, &vB};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float u0 = fmaf(T.b2, t3[q][2].x, fmaf(T.b1, t3[q][1].x, T.b0 * t3[q][0].x));
        const float u1 = fmaf(T.b2, t3[q][2].y, fmaf(T.b1, t3[q][1].y, T.b0 * t3[q][0].y));
        float o0, o1;
        asm("v_fma_f32 %[o0], %[a2], %[u0], %[cur0]\n\t"
            "v_fma_f32 %[o1], %[a2], %[u1], %[cur1]\n\t"
            "v_fmac_f32_e32 %[nxt0], %[a1], %[u0]\n\t"
            "v_fmac_f32_e32 %[nxt1], %[a1], %[u1]\n\t"
            "v_mul_f32_e32 %[cur0], %[a0], %[u0]\n\t"
            "v_mul_f32_e32 %[cur1], %[a0], %[u1]\n\t"
            "v_cvt_pk_bf16_f32 %[v], %[o0], %[o1]"
            : [o0] "=&v"(o0), [o1] "=&v"(o1), [v] "=v"(*v[q]), [cur0] "+v"(cur[q]->x), [nxt0] "+v"(nxt[q]->x), [cur1] "+v"(cur[q]->y),
              [nxt1] "+v"(nxt[q]->y)
            : [u0] "v"(u0), [u1] "v"(u1), [a0] LORA_BL_TAP(T.a0), [a1] LORA_BL_TAP(T.a1), [a2] LORA_BL_TAP(T.a2));
    }
#endif
}
// The same for a lane's first and last row behind the barrier: row 0 with the upper neighbour's last row above it (vu) and
// the wave's own rows 0 (t0, re-read from LDS) and 1 (t1, kept); row R - 1 as u = fma(b2, vd, part) with the lower
// neighbour's first row vd and part = fma(b1, t[R - 1], b0 * t[R - 2]) computed ahead of the barrier.
__device__ __forceinline__ void yz_edges(const f2 vu, const f2 t0, const f2 t1, const f2 vd, const f2 part, const TapsV &T, f2 &curA, f2 &nxtA,
                                         f2 &curB, f2 &nxtB, unsigned &vA, unsigned &vB) {
#if LORA_BL_PACKED
    f2 uA, uB, oA, oB;
    asm("v_pk_mul_f32 %[uA], %[b01], %[vu] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[uB], %[b2a0], %[vd], %[part] " LORA_LO "\n\t"
        "v_pk_fma_f32 %[uA], %[b01], %[t0], %[uA] " LORA_HI "\n\t"
        "v_pk_fma_f32 %[oB], %[a12], %[uB], %[curB] " LORA_HI "\n\t"
        "v_pk_fma_f32 %[uA], %[b2a0], %[t1], %[uA] " LORA_LO "\n\t"
        "v_pk_fma_f32 %[nxtB], %[a12], %[uB], %[nxtB] " LORA_LO "\n\t"
        "v_pk_fma_f32 %[oA], %[a12], %[uA], %[curA] " LORA_HI "\n\t"
        "v_pk_mul_f32 %[curB], %[b2a0], %[uB] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_fma_f32 %[nxtA], %[a12], %[uA], %[nxtA] " LORA_LO "\n\t"
        "v_pk_mul_f32 %[curA], %[b2a0], %[uA] op_sel:[1,0] op_sel_hi:[1,1]"
        : [uA] "=&v"(uA), [uB] "=&v"(uB), [oA] "=&v"(oA), [oB] "=&v"(oB), [curA] "+v"(curA), [nxtA] "+v"(nxtA), [curB] "+v"(curB), [nxtB] "+v"(nxtB)
        : [vu] "v"(vu), [t0] "v"(t0), [t1] "v"(t1), [vd] "v"(vd), [part] "v"(part), [b01] "v"(T.b01), [b2a0] "v"(T.b2a0), [a12] "v"(T.a12));
    vA = round_pair(oA);
    vB = round_pair(oB);
#else
    f2 *const cur[2] = {&curA, &curB}, *const nxt[2] = {&nxtA, &nxtB};
    unsigned *const v[2] = {&vA, &vB};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float u0 = q == 0 ? fmaf(T.b2, t1.x, fmaf(T.b1, t0.x, T.b0 * vu.x)) : fmaf(T.b2, vd.x, part.x);
        const float u1 = q == 0 ? fmaf(T.b2, t1.y, fmaf(T.b1, t0.y, T.b0 * vu.y)) : fmaf(T.b2, vd.y, part.y);
        float o0, o1;
        asm("v_fma_f32 %[o0], %[a2], %[u0], %[cur0]\n\t"
            "v_fma_f32 %[o1], %[a2], %[u1], %[cur1]\n\t"
            "v_fmac_f32_e32 %[nxt0], %[a1], %[u0]\n\t"
            "v_fmac_f32_e32 %[nxt1], %[a1], %[u1]\n\t"
            "v_mul_f32_e32 %[cur0], %[a0], %[u0]\n\t"
            "v_mul_f32_e32 %[cur1], %[a0], %[u1]\n\t"
            "v_cvt_pk_bf16_f32 %[v], %[o0], %[o1]"
            : [o0] "=&v"(o0), [o1] "=&v"(o1), [v] "=v"(*v[q]), [cur0] "+v"(cur[q]->x), [nxt0] "+v"(nxt[q]->x), [cur1] "+v"(cur[q]->y),
              [nxt1] "+v"(nxt[q]->y)
            : [u0] "v"(u0), [u1] "v"(u1), [a0] LORA_BL_TAP(T.a0), [a1] LORA_BL_TAP(T.a1), [a2] LORA_BL_TAP(T.a2));
    }
#endif
}
// part = fma(b1, t[R - 1], b0 * t[R - 2]): what row R - 1's y-pass can do ahead of the barrier
__device__ __forceinline__ f2 ypart(const f2 tm, const f2 t0, const TapsV &T) {
#if LORA_BL_PACKED
    f2 u;
    asm("v_pk_mul_f32 %[u], %[b01], %[tm] op_sel_hi:[0,1]\n\t"
        "s_nop 0\n\t"
        "v_pk_fma_f32 %[u], %[b01], %[t0], %[u] " LORA_HI
        : [u] "=&v"(u)
        : [tm] "v"(tm), [t0] "v"(t0), [b01] "v"(T.b01));
    return u;
#else
    return (f2){fmaf(T.b1, t0.x, T.b0 * tm.x), fmaf(T.b1, t0.y, T.b0 * tm.y)};
#endif
}
#undef LORA_LO
#undef LORA_HI

// Step p of a chunk whose first output plane is k0 takes input plane zin = k0 - K + p.  Level L (1 .. K) consumes the
// plane level L - 1 completed in the step before and completes its own plane zin - 2 L + 1; level K is the output: plane
// k0 + p - 3 K + 1, stored at the start of step p + 1.
template <int K, int NW>
__global__ __launch_bounds__(NW * 64, 4) void stencil3d_bf16_lanes_kernel(const ArgsBL a, const TapsSep W) {
    constexpr int R = 4;            // rows of a lane = rows of one LDS-DMA piece (4 x 256 bytes)
    constexpr int TH = R * NW;      // rows of a tile
    constexpr int OH = TH - 2 * K;  // output rows of a tile
    static_assert(K == 2 || K == 4, "an even number of applications (fused launches start at even steps)");
    // x-passed first / last row of every wave and level (written ahead of a half-step's barrier, read behind it)
    // Wave w keeps its rows in slot w + 1 of NW + 2, so that the upper neighbour's last row, the wave's own two rows and the
    // lower neighbour's first row are at fixed offsets 0 / 128 / 256 / 384 floats from ONE per-lane address (slots 0 and
    // NW + 1 are never written: what the top and the bottom wave read there feeds rows that are never valid).
    __shared__ __attribute__((aligned(16))) float edge_rows[K][NW + 2][2][kTileW];
    constexpr int kLevelStride = (NW + 2) * 2 * kTileW;  // floats
    // input planes: per wave a private ring of 1 KiB pieces (its four rows x 256 bytes of a plane) -- the plane on its way
    // in, the plane of the step and, for K = 4, the three before it: an EDGE step forces level-2 cells outside the interior
    // to the input's own values of plane zin - 3, and takes them from here.  (It used to load them from memory in the
    // middle of the step that needs them at its end: half a step is less than a round trip under load, and a rim tile's
    // steps took 11 % longer for 4 % more instructions -- the busiest CU of a launch runs a rim chunk and an inner one.)
    constexpr int D = K > 2 ? 5 : 2;
    __shared__ __attribute__((aligned(16))) unsigned ring[D][NW][R][kTileW / 2];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lin = xcd_contiguous(blockIdx.x, gridDim.x);
    const int TX = a.tiles_x, TY = a.tiles_y;
    // Spans (a.zc == 0, spans.h): the workgroup owns a range of the line of all (tile, plane) pairs and runs one SEGMENT per
    // tile the range touches; chunks (option fused_z_chunk / spans3 = 0): one segment, chunk `lin / tiles` of its tile.
    unsigned v0 = 0, v1 = 0;
    int team_col = 0;  // team spans: this workgroup's tile column
    if (a.zc == 0 && a.team) {
        // workgroups 0 .. TX team_ni - 1: piece lin / TX of column lin mod TX (x-neighbours side by side: one L2); the rest:
        // the extra pieces of the two rim columns
        int j;
        if (lin < a.team * a.team_ni) {
            j = lin / a.team;
            team_col = lin - j * a.team;
        } else {
            const int r = lin - a.team * a.team_ni;
            j = a.team_ni + (r >> 1);
            team_col = (r & 1) ? a.team - 1 : 0;
        }
        const bool rim_col = a.team_nr != a.team_ni && (team_col == 0 || team_col == a.team - 1);
        if (rim_col)
            span_range(a.sp2, j, v0, v1);
        else
            span_range(a.sp, j, v0, v1);
    } else if (a.zc == 0) {
        span_range(a.sp, lin, v0, v1);
    }
    const Spans &tsp = (a.team && a.team_nr != a.team_ni && (team_col == 0 || team_col == a.team - 1)) ? a.sp2 : a.sp;
    for (bool more = true, first = true; more; first = false) {
    int tx, ty, k0, zc;
    if (a.zc == 0) {
        int z0;
        bool any;
        if (a.team) {  // the line runs over tile ROWS; this workgroup is column lin mod TX of its team's row
            int t0;
            any = span_next(tsp, 3 * K - 1, v0, v1, 1, TY, t0, ty, z0, zc);
            ty = column_line_row(ty, TY);
            tx = team_col;
        } else {
            any = span_next(a.sp, 3 * K - 1, v0, v1, TX, TY, tx, ty, z0, zc);
        }
        more = v0 < v1;
        if (!any) continue;
        k0 = a.z_begin + z0;
    } else {
        int chunk;
        const int c0 = (a.z_end - a.z_begin + a.zc - 1) / a.zc, c1 = (a.z_end2 - a.z_begin2 + a.zc - 1) / a.zc;
        chunk_of(lin, c0 + c1, TX, TY, chunk, tx, ty);
        const int zb = chunk < c0 ? a.z_begin : a.z_begin2, ze = chunk < c0 ? a.z_end : a.z_end2;
        k0 = zb + (chunk < c0 ? chunk : chunk - c0) * a.zc;
        zc = min(a.zc, ze - k0);
        more = false;
    }
    // (the segment before is done with the rows in LDS when its slowest wave is)
    if (!first) __builtin_amdgcn_s_barrier();
#ifdef LORA_BL_TIMELINE
    if (a.timeline && threadIdx.x == 0) {
        if (a.timeline[4 * blockIdx.x + 0] == 0) a.timeline[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        a.timeline[4 * blockIdx.x + 2] += zc + 3 * K - 1;
        a.timeline[4 * blockIdx.x + 3] += 1;
    }
#endif
    const int X0 = tx * kOutW - 4, Y0 = ty * OH - K;  // interior coordinates of the tile's first column / row
    // this lane's cells: rows Y0 + R wv + r (r = 0 .. R - 1), columns X0 + 2 lane, + 1; padded: + 2 rows, + 4 columns,
    // clamped into the padded array (clamped cells only feed cells outside the interior, which EDGE forces, or nothing)
    const int col = X0 + 2 * lane;
    unsigned rowoff[R];  // uniform: padded row x ld, in bytes
#pragma unroll
    for (int r = 0; r < R; ++r) rowoff[r] = 2u * (unsigned) (min(max(Y0 + R * wv + r + 2, 0), a.m + 3) * a.ld);
    float *const edge_base = &edge_rows[0][wv][1][2 * lane];  // the upper neighbour's last row (slot wv), level 0
    const unsigned plane_bytes = 2u * (unsigned) a.plane;
    // the DMA piece of this lane: row lane / 16 of the wave's four, 16-byte chunk lane % 16 of the tile's 256-byte row
    // (padded column X0 + 4 + 8 (lane % 16): a multiple of 8 elements; chunks beyond the padded row re-read its last one)
    const unsigned dma_off = 2u * (unsigned) (min(max(Y0 + R * wv + (lane >> 4) + 2, 0), a.m + 3) * a.ld +
                                              min(X0 + 4 + 8 * (lane & 15), a.n));

    // stores: lanes 2 .. 61 write interior columns col, col + 1 of rows K .. TH - K - 1 of the tile (n is a multiple of 8:
    // a pair never straddles the end of a row)
    // (byte offset inside a padded row; the row itself is the scalar offset of the store; out of range = no store)
    const unsigned st_off = (lane >= 2 && lane < 62 && col >= 0 && col < a.n) ? 2u * (unsigned) (col + 4) : 0x80000000u;
    bool st_row[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int trow = R * wv + r;
        st_row[r] = trow >= K && trow < TH - K && Y0 + trow < a.m;
    }

    // EDGE: which of this lane's cells are interior cells -- in x as a mask over the pair's two halves, in y per row
    const bool xy_rim = X0 < 0 || X0 + kTileW > a.n || Y0 < 0 || Y0 + TH > a.m;  // (uniform over the workgroup)
    const unsigned colmask = ((unsigned) col < (unsigned) a.n ? 0x0000ffffu : 0u) | ((unsigned) (col + 1) < (unsigned) a.n ? 0xffff0000u : 0u);
    bool row_in[R];
#pragma unroll
    for (int r = 0; r < R; ++r) row_in[r] = (unsigned) (Y0 + R * wv + r) < (unsigned) a.m;

    // Per level and point two fp32 sums, slots 0 / 1; in a step of phase P (= p mod 2, a compile-time constant of the
    // step's copy) acc[l][P] holds the plane that completes (dz = 0, 1 taps in) and acc[l][1 - P] the one above it (dz = 0).
    f2 acc[K][2][R];  // (a lane's two columns side by side: an aligned register pair, what the packed instructions take)
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[l][q][r] = (f2){0.0f, 0.0f};
    unsigned done[K][R];  // done[l], l >= 1: the plane level l completed in the step before (packed bf16 pairs)
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int r = 0; r < R; ++r) done[l][r] = 0u;
    unsigned outp[R];  // the output plane (level K) a step completes: stored in the middle of the step

    // (Plane addresses as 32 x 32 -> 64-bit products of a plane index and plane_bytes: two scalar instructions; a plane's
    // descriptor covers the whole plane and a row is picked by the scalar offset of the access -- per-row descriptors and
    // 64 x 64-bit address arithmetic were 65 scalar instructions per step, a third of them for the four stores.)
    auto issue_plane = [&](int p, int slot) {  // this wave's four rows of input plane p of the chunk -> ring[slot][wv]
        const unsigned z = (unsigned) min(max(k0 - K + p + 1, 0), a.h + 1);
        const char *src = reinterpret_cast<const char *>(a.in) + (unsigned long long) z * plane_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src), 0, plane_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *) &ring[slot][wv][0][0], 16, dma_off, 0, 0, 0);
    };
    unsigned st_bytes[R];  // what a row's descriptor lets through: the plane, or nothing for rows this wave does not store
#pragma unroll
    for (int r = 0; r < R; ++r) st_bytes[r] = st_row[r] ? plane_bytes : 0u;
    auto store_plane = [&](int p_done) {  // the output plane completed in step p_done
        const int o = p_done - 3 * K + 1;
        const bool live = (unsigned) o < (unsigned) zc;
        char *const dst = reinterpret_cast<char *>(a.out) + (unsigned long long) (unsigned) (k0 + max(o, 0) + 1) * plane_bytes;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, live ? st_bytes[r] : 0u, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(outp[r], rs, st_off, rowoff[r], 0);
        }
    };
    issue_plane(0, 0);
    int slot = 0;  // the ring slot of the step's input plane: step p's plane lives in slot p mod D
    // LDS byte address of this lane's dword of its wave's piece in slot 0 (slot 1: + 16 KB, row r: + 256 bytes)
    static_assert(sizeof(ring[0]) == NW * 4 * 256, "the slot stride");
    constexpr unsigned kSlotBytes = NW * 4 * 256;
    const unsigned ring_addr = (unsigned) (size_t) (__attribute__((address_space(3))) void *) &ring[0][wv][0][lane];
    // The nine taps live in VECTOR registers: on gfx950 an fp32 multiply-add with a scalar (constant-bus) operand issues at
    // HALF the rate of the all-VGPR form -- v_fmac_f32 acc += s * v: 2.9 cycles of SIMD time per wave instruction at four
    // waves per SIMD, acc += v * v: 1.3 - 1.9; v_mul_f32 3.3 against 1.35 (profiles/r04_fp32_rate_probe.txt)
    TapsV T = {W.c[0], W.c[1], W.c[2], W.b[0], W.b[1], W.b[2], W.a[0], W.a[1], W.a[2], {W.b[0], W.b[1]}, {W.b[2], W.a[0]}, {W.a[1], W.a[2]}};
#if LORA_BL_ABLATE & 32
    asm volatile("" : "+v"(T.c0), "+v"(T.c2));  // (DPP operands are vector registers whatever the rest is)
#elif LORA_BL_PACKED
    asm volatile("" : "+v"(T.c0), "+v"(T.c1), "+v"(T.c2), "+v"(T.b01), "+v"(T.b2a0), "+v"(T.a12));
#else
    asm volatile("" : "+v"(T.c0), "+v"(T.c1), "+v"(T.c2), "+v"(T.b0), "+v"(T.b1), "+v"(T.b2), "+v"(T.a0), "+v"(T.a1), "+v"(T.a2));
#endif

#if LORA_BL_STAMP
    long long stamp_sum = 0, stamp_t0 = 0;
    int stamp_n = 0;
#define LORA_BL_T0(n)                                                   \
    if constexpr (LORA_BL_STAMP == (n)) {                               \
        __builtin_amdgcn_sched_barrier(0);                              \
        stamp_t0 = __builtin_amdgcn_s_memtime();                        \
        __builtin_amdgcn_sched_barrier(0);                              \
    }
#define LORA_BL_T1(n)                                                   \
    if constexpr (LORA_BL_STAMP == (n)) {                               \
        __builtin_amdgcn_sched_barrier(0);                              \
        stamp_sum += __builtin_amdgcn_s_memtime() - stamp_t0;           \
        stamp_n += 1;                                                   \
        __builtin_amdgcn_sched_barrier(0);                              \
    }
#else
#define LORA_BL_T0(n)
#define LORA_BL_T1(n)
#endif
    auto step = [&](const int p, auto phase_tag, auto edge_tag, auto fill_tag) {
        constexpr int P = decltype(phase_tag)::value, Q = 1 - P;
        constexpr bool EDGE = decltype(edge_tag)::value;
        // FILL: the copy of the step for a segment's first and last steps, where some levels have nothing to do yet / any
        // more.  Level index l (levels l -> l + 1) takes in plane zin - 2 l, which feeds planes zin - 2 l - 1 .. + 1 of
        // level l + 1; the segment needs planes k0 - (K - l - 1) .. k0 + zc - 1 + (K - l - 1) of that level: the level has
        // work in steps 3 l .. zc + 2 K + l - 1 only.  Of a segment's zc + 3 K - 1 steps that is 6 steps' worth of
        // arithmetic in 24 level-steps (K = 4) -- 3.6 % of a 154-plane chunk.
        constexpr bool FILL = decltype(fill_tag)::value;
        auto active = [&](int l) { return !FILL || (p >= 3 * l && p <= zc + 2 * K + l - 1); };
        const int zin = k0 - K + p;
        unsigned hv[R];     // EDGE: what level-2 cells outside the interior are forced to
        f2 tk[K][2];  // per level: x-passed row 1, and row R - 1's y-pass short of the lower neighbour's row
        unsigned raw[R];  // the input plane's dwords of this lane
        auto pickup = [&]() {  // this lane's dwords of the step's input plane (and an EDGE step's halo values) out of the ring
            // Picked up here, not at the top of the step: four registers less while the upper levels run.  Issue and wait
            // in ONE statement: split in two (the read a level earlier, to hide its round trip) the compiler is free to
            // copy the destination registers before the data has arrived -- wrong planes, seen.  And by hand at all:
            // the compiler knows that the DMA writes `ring` but not WHICH slot, so a plain read of slot P waits
            // (s_waitcnt vmcnt) for the DMA of slot Q issued a moment ago -- the whole latency of the prefetch, every
            // step.  Plane p landed before the step's own vmcnt(0) above.
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 lo, hi;
            const unsigned cur = ring_addr + (unsigned) slot * kSlotBytes;
            if constexpr (EDGE && K > 2) {
                // ... and with it the lane's dwords of input plane p - 3 (level 2 completes plane zin - 3 in this step; its
                // cells outside the interior take the input's values there)
                const int sh = slot + 2 >= D ? slot + 2 - D : slot + 2;
                const unsigned old = ring_addr + (unsigned) sh * kSlotBytes;
                u32x2 hlo, hhi;
                asm volatile("ds_read2st64_b32 %0, %4 offset1:1\n\t"
                             "ds_read2st64_b32 %1, %4 offset0:2 offset1:3\n\t"
                             "ds_read2st64_b32 %2, %5 offset1:1\n\t"
                             "ds_read2st64_b32 %3, %5 offset0:2 offset1:3\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(lo), "=&v"(hi), "=&v"(hlo), "=&v"(hhi)
                             : "v"(cur), "v"(old)
                             : "memory");
                hv[0] = hlo.x;
                hv[1] = hlo.y;
                hv[2] = hhi.x;
                hv[3] = hhi.y;
            } else {
                asm volatile("ds_read2st64_b32 %0, %2 offset1:1\n\t"
                             "ds_read2st64_b32 %1, %2 offset0:2 offset1:3\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(lo), "=&v"(hi)
                             : "v"(cur)
                             : "memory");
            }
            raw[0] = lo.x;
            raw[1] = lo.y;
            raw[2] = hi.x;
            raw[3] = hi.y;
        };
        // level l + 1 from the plane of level l (l = 0: the input plane): everything that needs no other wave's row
        auto ahead = [&](auto level_tag) {
            constexpr int l = decltype(level_tag)::value;
            if constexpr (l == 0) pickup();
            const unsigned(&in)[R] = l == 0 ? raw : done[l];
            f2 t[R];
            xpass_rows(in, T.c0, T.c1, T.c2, t);
            *reinterpret_cast<f2 *>(edge_base + l * kLevelStride + 128) = t[0];
            *reinterpret_cast<f2 *>(edge_base + l * kLevelStride + 256) = t[R - 1];
            {
                static_assert(R == 4, "rows 1 and 2 are the two inner rows");
                unsigned v1, v2;
                yz_rows2(t[0], t[1], t[2], t[1], t[2], t[3], T, acc[l][P][1], acc[l][Q][1], acc[l][P][2], acc[l][Q][2], v1, v2);
                if constexpr (l == K - 1) {
                    outp[1] = v1;
                    outp[2] = v2;
                } else {
                    done[l + 1][1] = v1;
                    done[l + 1][2] = v2;
                }
            }
            tk[l][0] = t[1];
            tk[l][1] = ypart(t[R - 2], t[R - 1], T);
            // one level at a time: eight points are parallelism enough, and a scheduler that interleaves levels keeps
            // several sets of x-passed rows alive (the kernel then spills at its 128 registers)
            __builtin_amdgcn_sched_barrier(0);
        };
        // ... and behind the barrier rows 0 and R - 1, with the neighbours' rows (and the wave's own first row again: cheaper
        // than two more registers per level held across the barrier)
        // The three rows a level needs from LDS behind the barrier: the upper neighbour's last row, the wave's own first
        // row, the lower neighbour's first row.  A half-step's reads are all issued before the first is used: the second
        // level's rows travel while the first level computes (one LDS round trip per half-step on a wave's critical path,
        // not one per level).
        struct Rows3 {
            f2 vu, t0, vd;
        };
        auto fetch = [&](auto level_tag) {
            constexpr int l = decltype(level_tag)::value;
            Rows3 q;
            q.vu = *reinterpret_cast<const f2 *>(edge_base + l * kLevelStride);
            q.t0 = *reinterpret_cast<const f2 *>(edge_base + l * kLevelStride + 128);
            q.vd = *reinterpret_cast<const f2 *>(edge_base + l * kLevelStride + 384);
            return q;
        };
        auto behind = [&](auto level_tag, const Rows3 &q) {
            constexpr int l = decltype(level_tag)::value;
            unsigned v0, v3;
            yz_edges(q.vu, q.t0, tk[l][0], q.vd, tk[l][1], T, acc[l][P][0], acc[l][Q][0], acc[l][P][R - 1], acc[l][Q][R - 1], v0, v3);
            if constexpr (l == K - 1) {
                outp[0] = v0;
                outp[R - 1] = v3;
            } else {
                done[l + 1][0] = v0;
                done[l + 1][R - 1] = v3;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // The step runs the levels top down (level l + 1 has consumed done[l + 1] when level l overwrites it) in two halves
        // with a barrier each: the x-passed rows a wave keeps for behind the barrier are then two levels' worth, not four, and
        // one copy of edge_rows serves (a half's rows are rewritten a whole step later, two barriers on).
        LORA_BL_T0(1);
        if constexpr (FILL) {  // (tests uniform over the workgroup; every wave meets the step's two barriers)
            if constexpr (K == 4) {
                if (active(3)) ahead(std::integral_constant<int, 3>{});
                if (active(2)) ahead(std::integral_constant<int, 2>{});
            } else {
                if (active(1)) ahead(std::integral_constant<int, 1>{});
            }
        } else if constexpr (K == 4) {
            ahead(std::integral_constant<int, 3>{});
            ahead(std::integral_constant<int, 2>{});
        } else {
            ahead(std::integral_constant<int, 1>{});
        }
        LORA_BL_T1(1);
        LORA_BL_T0(2);
        if constexpr (LORA_BL_ABLATE & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LORA_BL_T1(2);
        LORA_BL_T0(3);
        if constexpr (FILL) {
            if constexpr (K == 4) {
                if (active(3)) behind(std::integral_constant<int, 3>{}, fetch(std::integral_constant<int, 3>{}));
                if (active(2)) behind(std::integral_constant<int, 2>{}, fetch(std::integral_constant<int, 2>{}));
            } else {
                if (active(1)) behind(std::integral_constant<int, 1>{}, fetch(std::integral_constant<int, 1>{}));
            }
        } else if constexpr (K == 4) {
            const Rows3 q3 = fetch(std::integral_constant<int, 3>{}), q2 = fetch(std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
            behind(std::integral_constant<int, 3>{}, q3);
            behind(std::integral_constant<int, 2>{}, q2);
        } else {
            behind(std::integral_constant<int, 1>{}, fetch(std::integral_constant<int, 1>{}));
        }
        LORA_BL_T1(3);
        LORA_BL_T0(4);
        // The output plane of this step is complete.  Everything issued at this point of the step before has had a whole
        // step to complete -- the DMA of plane p, which the lower half is about to read, and the stores of the plane before
        // (and an EDGE step's halo loads, which it waited for itself) -- so the one wait of the step costs nothing, and no
        // count relies on the order in which loads and stores complete.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(LORA_BL_ABLATE & 2)) store_plane(p);
        const int slot_next = slot + 1 == D ? 0 : slot + 1;
        if constexpr (!(LORA_BL_ABLATE & 4)) issue_plane(p + 1, slot_next);
        LORA_BL_T1(4);
        LORA_BL_T0(5);
        if constexpr (FILL) {
            if constexpr (K == 4)
                if (active(1)) ahead(std::integral_constant<int, 1>{});
            if (active(0))
                ahead(std::integral_constant<int, 0>{});
            else
                pickup();  // (the halo values the forcing below takes)
        } else if constexpr (K == 4) {
            ahead(std::integral_constant<int, 1>{});
            ahead(std::integral_constant<int, 0>{});
        } else {
            ahead(std::integral_constant<int, 0>{});
        }
        LORA_BL_T1(5);
        LORA_BL_T0(6);
        if constexpr (LORA_BL_ABLATE & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LORA_BL_T1(6);
        LORA_BL_T0(7);
        if constexpr (FILL) {
            if constexpr (K == 4)
                if (active(1)) behind(std::integral_constant<int, 1>{}, fetch(std::integral_constant<int, 1>{}));
            if (active(0)) behind(std::integral_constant<int, 0>{}, fetch(std::integral_constant<int, 0>{}));
        } else if constexpr (K == 4) {
            const Rows3 q1 = fetch(std::integral_constant<int, 1>{}), q0 = fetch(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            behind(std::integral_constant<int, 1>{}, q1);
            behind(std::integral_constant<int, 0>{}, q0);
        } else {
            behind(std::integral_constant<int, 0>{}, fetch(std::integral_constant<int, 0>{}));
        }
        LORA_BL_T1(7);
        if constexpr (EDGE) {
            // the plane level L just completed, zin - 2 L + 1: its cells outside the interior are forced -- to 0 at odd
            // levels, at level 2 to the source buffer's own value there
#pragma unroll
            for (int L = 1; L < K; ++L) {
                const int z = zin - 2 * L + 1;
                const bool z_in = (unsigned) z < (unsigned) a.h;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned keep = (z_in && row_in[r]) ? colmask : 0u;
                    done[L][r] = L % 2 == 0 ? ((done[L][r] & keep) | (hv[r] & ~keep)) : (done[L][r] & keep);
                }
            }
        }
        slot = slot_next;
        __builtin_amdgcn_sched_barrier(0);
    };

    // Steps whose intermediate planes can lie outside the z range: level L's plane zin - 2 L + 1 is below plane 0 while
    // p < 3 K - 3 - k0 (L = K - 1) and beyond plane h - 1 from p = h - k0 + K + 1 on (L = 1).  Tiles on the rim in x / y run
    // the EDGE copy throughout.  Steps come in turns of two (one per phase); the EDGE ranges are widened to whole turns, and
    // a chunk may run one step past its last plane (loads clamped, stores switched off).
    const int steps = zc + 3 * K - 1, turns = (steps + 1) / 2;
    int p1 = min(max(3 * K - 3 - k0, 0), steps), p2 = min(max(a.h - k0 + K + 1, p1), steps);
    if (xy_rim) p1 = steps;
    if (LORA_BL_ABLATE & 8) p1 = 0, p2 = steps;
    if (LORA_BL_ABLATE & 16) p1 = steps;
    const int t1 = min((p1 + 1) / 2, turns), t2 = min(max(p2 / 2, t1), turns);
    // ... and the first 3 (K - 1) steps and the steps from zc + 2 K on run the FILL copy (an EDGE copy: forcing cells that
    // need none changes nothing).
    const int tf = (LORA_BL_ABLATE & 256) ? 0 : min((3 * (K - 1) + 1) / 2, turns);
    const int td = (LORA_BL_ABLATE & 256) ? turns : max(min((zc + 2 * K) / 2, turns), tf);
    auto turn = [&](const int p, auto edge_tag, auto fill_tag) {
        step(p, std::integral_constant<int, 0>{}, edge_tag, fill_tag);
        step(p + 1, std::integral_constant<int, 1>{}, edge_tag, fill_tag);
    };
    int t = 0;
    for (; t < tf; ++t) turn(2 * t, std::true_type{}, std::true_type{});
    for (; t < min(t1, td); ++t) turn(2 * t, std::true_type{}, std::false_type{});
    for (; t < min(t2, td); ++t) turn(2 * t, std::false_type{}, std::false_type{});
    for (; t < td; ++t) turn(2 * t, std::true_type{}, std::false_type{});
    for (; t < turns; ++t) turn(2 * t, std::true_type{}, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the DMA of a plane nobody reads: it must not land in a dead workgroup's LDS,
                                                      //  nor in the ring after the first plane of the next segment)
#ifdef LORA_BL_TIMELINE
    if (a.timeline && threadIdx.x == 0) a.timeline[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
#if LORA_BL_STAMP
    if (lane == 0) {  // (of the workgroup's last segment)
        a.stamps[((long) blockIdx.x * NW + wv) * 2] = stamp_sum;
        a.stamps[((long) blockIdx.x * NW + wv) * 2 + 1] = ((long long) xy_rim << 32) | stamp_n;
    }
#endif
    }  // segments
}
#undef LORA_BL_T0
#undef LORA_BL_T1

template <int K, int NW>
hipError_t launch_t(const Plan &p, const void *in, void *out, int begin, int end, int begin2, int end2, hipStream_t s) {
    constexpr int OH = 4 * NW - 2 * K;
    auto kernel = stencil3d_bf16_lanes_kernel<K, NW>;
    static int per_cu[64] = {0};  // resolved once per device (and with it the kernel itself: lora_plan_create's share)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (per_cu[dev] == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NW * 64, 0) != hipSuccess || nb < 1) {
            (void) hipGetLastError();
            nb = -1;  // no workgroup of this kernel fits a CU of this device (it wants the whole 160 KB of LDS)
        }
        per_cu[dev] = nb;
    }
    if (per_cu[dev] < 0) return hipErrorLaunchOutOfResources;
    if (end <= begin && end2 <= begin2) return hipSuccess;  // prepare_3d_bf16_lanes()
    ArgsBL a{};
    a.in = static_cast<const u16 *>(in);
    a.out = static_cast<u16 *>(out);
    a.h = p.dims[0];
    a.m = p.dims[1];
    a.n = p.dims[2];
    a.ld = a.n + 8;
    a.plane = (long) (a.m + 4) * (a.n + 8);
    if (a.plane * 2 >= (1L << 31)) return hipErrorInvalidValue;  // 32-bit byte offsets inside a plane
    a.z_begin = begin;
    a.z_end = end;
    a.tiles_x = (a.n + kOutW - 1) / kOutW;
    a.tiles_y = (a.m + OH - 1) / OH;
    const long tiles = (long) a.tiles_x * a.tiles_y;
    // How the launch is cut along z (spans.h): chunks of option fused_z_chunk; else spans (option spans3 = 1, or by itself
    // where they pay: spans_pay); else equal chunks by the model of rounds of workgroups.
    const long slots = (long) std::max(per_cu[dev], 1) * cus, depth = end - begin, depth2 = std::max(end2 - begin2, 0);
    constexpr int S = 3 * K - 1;
    long nblocks = 0;
    a.zc = 0;
    a.z_begin2 = begin2;
    a.z_end2 = begin2 + (int) depth2;
    if (p.fused_z_chunk > 0) {
        a.zc = std::min((long) p.fused_z_chunk, std::max(depth, depth2));
    } else if (depth2 > 0) {  // two ranges (a slab's end regions): chunks, as many workgroups as two ranges of the longer depth
        a.zc = chunk_model(2 * tiles, std::max(depth, depth2), S, slots, 4 * K, nullptr);
    } else {
        const int zc_model = chunk_model(tiles, depth, S, slots, 4 * K, nullptr);
        const bool spans = p.spans3 >= 1 || (p.spans3 < 0 && spans_pay(tiles, depth, S, slots, zc_model, 2.0 * (double) a.plane * (double) depth, 256.0e6));
        const bool teams = p.spans3 == 2 || (p.spans3 < 0 && !spans && teams_pay(a.tiles_x, a.tiles_y, depth, S, slots, zc_model));
        if (teams && a.tiles_x <= slots) {
            // TEAM spans: the line runs over tile rows and is cut into one piece per team of tiles_x workgroups, one per
            // tile of the row -- x-neighbours stay at the same depth (and, dealt out contiguously, in the same XCD's L2)
            // The rim columns' tiles run EDGE steps, ~5 % longer each (tools/probes/bf16_lanes_timeline.hip: 2.47 against 2.34
            // us per step; with equal pieces the launch ended with the rim columns' workgroups at 775 us, the others at
            // 734).  So the rim columns are cut into more, shorter pieces than the columns between -- 38 and 36 at 768^3, which
            // also uses all 256 CUs instead of 252.  The columns between stay in lockstep with one another.
            long ni, nr;
            team_pieces(a.tiles_x, slots, !(LORA_BL_ABLATE & 512), &ni, &nr);
            // (and within a column between them the first and last tile ROW are rim tiles too: their planes weigh 19 : 18)
            const bool weigh = a.tiles_x >= 3 && !(LORA_BL_ABLATE & 512);
            const long ti = ni > 0 ? spans_setup_column(a.sp, a.tiles_y, depth, S, ni, weigh ? 19 : 1, weigh ? 18 : 1) : 0;
            const long tr = nr > 0 ? spans_setup_column(a.sp2, a.tiles_y, depth, S, nr, 1, 1) : 0;
            if (ti > 0 && tr >= ti) {
                a.team = a.tiles_x;
                a.team_ni = (int) ti;
                a.team_nr = (int) (a.tiles_x >= 3 ? tr : ti);
                nblocks = a.tiles_x >= 3 ? ti * (a.tiles_x - 2) + 2 * tr : ti * a.tiles_x;
            }
        } else if (spans) {
            nblocks = spans_setup(a.sp, a.tiles_x, a.tiles_y, depth, S, slots, 10, 9);
        }
        if (nblocks == 0) a.zc = zc_model;  // (or a line that does not fit 31 bits of cost units)
    }
    if (a.zc > 0) nblocks = tiles * ((depth + a.zc - 1) / a.zc + (depth2 + a.zc - 1) / a.zc);
    if (nblocks > 0x7fffffffL) return hipErrorInvalidValue;
    TapsSep w;
    for (int k = 0; k < 3; ++k) {
        w.c[k] = p.sep[k];
        w.b[k] = p.sep[3 + k];
        w.a[k] = p.sep[6 + k];
    }
#if LORA_BL_STAMP
    extern long long *g_bl_stamps;
    extern long g_bl_stamp_blocks;
    a.stamps = g_bl_stamps;
    g_bl_stamp_blocks = nblocks;
#endif
#ifdef LORA_BL_TIMELINE
    extern long long *g_bl_timeline;
    extern long g_bl_timeline_blocks;
    a.timeline = g_bl_timeline;
    g_bl_timeline_blocks = nblocks;
#endif
    hipLaunchKernelGGL(kernel, dim3((unsigned) nblocks), dim3(NW * 64), 0, s, a, w);
    return hipGetLastError();
}

}  // namespace

// K = 4 (or 2) applications in one launch over interior planes [begin, end); exactly separable box taps, reference boundary
// (and planes [begin2, end2) in the same launch: kernels_3d_lanes.hip)
hipError_t launch_3d_bf16_lanes(const Plan &p, int K, const void *in, void *out, int begin, int end, hipStream_t s, int begin2, int end2) {
    if (p.boundary != LORA_BC_REFERENCE || p.dtype != LORA_BF16 || p.tapset != TAPS3D_SEP) return hipErrorNotSupported;
#ifdef LORA_BL_NW  // (tools/probes/bf16_lanes_ablate.hip: another workgroup size)
    if (K == 4) return launch_t<4, LORA_BL_NW>(p, in, out, begin, end, begin2, end2, s);
    if (K == 2) return launch_t<2, LORA_BL_NW>(p, in, out, begin, end, begin2, end2, s);
#else
    if (K == 4) return launch_t<4, 16>(p, in, out, begin, end, begin2, end2, s);
    if (K == 2) return launch_t<2, 16>(p, in, out, begin, end, begin2, end2, s);
#endif
    return hipErrorInvalidValue;
}

// one-time host work (kernel resolution, residency query) of the plan's instantiations; no launch.  False when no
// workgroup of the kernel fits a CU of the current device (the plan then keeps the tile kernels).
bool prepare_3d_bf16_lanes(const Plan &p) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void) hipGetLastError();
        return true;  // no device: nothing to ask (plans made on a host without one only answer host-side queries)
    }
    return launch_3d_bf16_lanes(p, 4, nullptr, nullptr, 0, 0, nullptr) == hipSuccess &&
           launch_3d_bf16_lanes(p, 2, nullptr, nullptr, 0, 0, nullptr) == hipSuccess;
}

const char *kernel_name_3d_bf16_lanes(const Plan &) { return "stencil3d_bf16_lanes_kernel"; }

}  // namespace lora
