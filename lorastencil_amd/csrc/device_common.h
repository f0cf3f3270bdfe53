// device_common.h -- small device helpers shared by the 2D kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "engine.h"

namespace lora {

// Timing experiments that remove loads / stores from a kernel (plan option "ablate", results wrong by construction)
// exist only in builds made with -DLORA_DIAGNOSTICS; in the shipped library the expression folds to 0.
#ifdef LORA_DIAGNOSTICS
#define LORA_ABLATE(a) ((a).ablate)
#else
#define LORA_ABLATE(a) 0
#endif

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// Cache policy of the fused kernels' output stores (the `aux` operand of raw_buffer_store: 2 = nt on gfx950).  A launch
// never re-reads what it writes, and streamed lines leave the L2 / Infinity Cache to the rows and planes neighbouring
// workgroups re-read: 1.5 - 3 % per launch (old / new library in one process, two alternations: star3d1r 512^3 four sweeps
// 632, 624 -> 613, 613 us; box3d1r 768^3 2114, 2100 -> 2085, 2075; star2d1r 16384^2 six sweeps 1240, 1243 -> 1221, 1223;
// profiles/r03_nt_stores_ab.txt; run-to-run noise on one box is about +-1.5 %, so this is a small effect at best.  The bf16
// kernel's 8-byte stores got SLOWER with it, 567-574 -> 618-631 us, and keep the default policy).
constexpr int kStoreNT = 2;

// Which of the 49 taps a kernel instantiation evaluates (dy, dx in 0..6).
template <int TAPSET>
__host__ __device__ constexpr bool tap_on(int dy, int dx) {
    const int ay = dy < 3 ? 3 - dy : dy - 3;
    const int ax = dx < 3 ? 3 - dx : dx - 3;
    return TAPSET == TAPS2D_BOX ? true : (TAPSET == TAPS2D_STAR ? (ay == 0 || ax == 0) : (ay + ax <= 3));
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one, each XCD has its own L2).
// Give each XCD a contiguous run of the linear tile order (bijective for any block count).
__device__ __forceinline__ int xcd_contiguous(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, slot = b >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

// Linear index -> tile coordinates, panel-major: panels of `pw` tile columns, each walked row by row, so that
// a tile's horizontal neighbours and the tile row above are recent in the same L2.
__device__ __forceinline__ void panel_major(int lin, int tiles_x, int tiles_y, int pw, int &ty, int &tx) {
    const int per_panel = pw * tiles_y;
    const int full = tiles_x / pw;
    const int p = lin / per_panel;
    if (p < full) {
        const int q = lin - p * per_panel;
        ty = q / pw;
        tx = p * pw + (q - ty * pw);
    } else {
        const int rem = tiles_x - full * pw;
        const int q = lin - full * per_panel;
        ty = q / rem;
        tx = full * pw + (q - ty * rem);
    }
}

}  // namespace lora
