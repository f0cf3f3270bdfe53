// spans.h -- how a launch of the register-resident 3D kernels (kernels_3d_lanes.hip, kernels_3d_bf16_lanes.hip) is cut
// along z, when it is not cut into equal chunks per tile.
//
// Those kernels run ONE workgroup per CU; a workgroup that starts anywhere along z runs S steps (its pipeline filling)
// before its first output plane, and a launch takes as long as its busiest CU.  Equal chunks per tile leave CUs idle
// whenever tiles x chunks is not a multiple of the CU count (768^3 bf16: 98 tiles x 5 chunks = 490 workgroups of 165 steps
// on 256 CUs = 330 steps, where 98 x 768 / 256 + 11 = 305 would do).  SPANS put all (tile, plane) pairs on one line --
// tile-major, tiles in rim-first order, a step of a rim tile weighted by what its EDGE steps cost, S steps of start per
// tile -- and cut the line into one equal piece per resident workgroup; a workgroup runs the one or two (on shallow regions
// several) tile SEGMENTS its piece touches.
//
// What it buys is measured, not assumed (tools/spans_sweep.py, profiles/r04_spans_*): fewer cycles everywhere, but
// neighbouring tiles are no longer at the same depth at the same time, so the rows they share -- and the 128-byte lines
// their misaligned row pieces straddle -- are fetched from HBM twice.  TEAM spans (the line over tile rows, a piece per
// team of a row's workgroups) keep the neighbours in x together.  Plan option spans3: -1 by the kernels' rules (spans_pay,
// teams_pay), 0 chunks, 1 spans, 2 team spans.  Chunked launches have two layouts of their own in kernels_3d_lanes.hip:
// chunks dealt to resident workgroups when there are more chunks than CUs, and one chunk more for the slow rim tiles when a
// single round has CUs to spare.
#pragma once

#include <algorithm>

namespace lora {

struct Spans {
    int q, r;        // a workgroup's share of the line's cost units: q, the first r workgroups one more
    int vt;          // steps of a whole tile: the region's planes + the S steps of a start
    int wrim, win;   // cost units of a step of a tile on the rim of the grid (EDGE steps throughout) / of an inner tile
    int nrim;        // tiles on the rim (they come first on the line)
};

// Host: fill `sp` for a region of `depth` planes of tiles_x x tiles_y tiles on `slots` resident workgroups; returns the
// number of workgroups to launch, or 0 when the line does not fit 31 bits of cost units (then: chunks).
inline long spans_setup(Spans &sp, int tiles_x, int tiles_y, long depth, int S, long slots, int wrim, int win) {
    const long tiles = (long) tiles_x * tiles_y;
    const long nrim = (tiles_x < 3 || tiles_y < 3) ? tiles : 2L * tiles_x + 2L * (tiles_y - 2);
    sp.vt = (int) (depth + S);
    sp.wrim = wrim;
    sp.win = win;
    sp.nrim = (int) nrim;
    const long units = (depth + S) * (nrim * wrim + (tiles - nrim) * win);
    if (units >= (1L << 31)) return 0;
    // no more workgroups than pieces worth a start (a piece of at least ~S steps)
    const long wgs = std::max(1L, std::min(slots, units / ((long) S * win)));
    sp.q = (int) (units / wgs);
    sp.r = (int) (units % wgs);
    return wgs;
}

// The line of ONE tile column (team spans): tiles_y tiles, the first and last of them (rim rows: EDGE steps) at weight wrim and
// FIRST on the line, the others at win.  Tile t of this line is row column_line_row(t, tiles_y).
inline long spans_setup_column(Spans &sp, int tiles_y, long depth, int S, long slots, int wrim, int win) {
    const long nrim = tiles_y >= 3 ? 2 : tiles_y;
    sp.vt = (int) (depth + S);
    sp.wrim = wrim;
    sp.win = win;
    sp.nrim = (int) nrim;
    const long units = (depth + S) * (nrim * wrim + (tiles_y - nrim) * win);
    if (units >= (1L << 31)) return 0;
    const long wgs = std::max(1L, std::min(slots, units / ((long) S * win)));
    sp.q = (int) (units / wgs);
    sp.r = (int) (units % wgs);
    return wgs;
}

// Team spans with the rim columns cut finer: how many pieces a tile column between the rim columns gets (ni) and how many a
// rim column (nr >= ni), so that (tiles_x - 2) ni + 2 nr <= slots and nr / ni is about what a rim tile's steps cost more
// (EDGE steps: ~5.5 %).  768^3 bf16 (7 columns, 256 CUs): 36 and 38.
inline void team_pieces(int tiles_x, long slots, bool finer_rims, long *ni, long *nr) {
    *ni = *nr = slots / tiles_x;
    if (tiles_x >= 3 && finer_rims) {
        *ni = (long) ((double) slots / ((double) (tiles_x - 2) + 2.0 * 1.055));
        *nr = *ni > 0 ? std::min((slots - (tiles_x - 2) * *ni) / 2, (long) ((double) *ni * 1.07) + 1) : 0;
    }
}

// Equal chunks per tile: the chunk length by a model of rounds of workgroups.  A chunk runs S steps beyond its own planes,
// so chunks should be long; a round of workgroups (one per slot) takes its steps whatever its kernels do, and a LAST,
// partly filled round is cheaper than a full one only down to about two thirds of it.  Fitted to a sweep of chunk lengths
// (star3d1r 512^3 fp64, 110 tiles on 256 CUs, tools/thin3d_check.py; x = workgroups / CUs, time / 2.4 us / steps per chunk:
// x = 0.86 -> 1.00, 1.29 -> 1.78, 1.72 -> 1.99, 2.15 -> 2.70, 2.58 -> 2.97, 3.008 -> 3.65, 3.44 -> 3.97, 3.87 -> 4.16,
// 4.30 -> 4.88, 5.16 -> 5.8, 6.9 -> 7.2):   rounds_eff(x) = 1.03 floor(x) + (0.65 + 0.4 frac(x) if frac(x) > 0)
// and the chunk count that minimises rounds_eff x (zc + S) wins (chunks of at least `min_chunk` planes while the depth
// allows).  `cost`: that minimum, in steps.
inline int chunk_model(long tiles, long depth, int S, long slots, long min_chunk, double *cost) {
    double best = 0.0;
    long best_c = 1;
    for (long c = 1; c <= std::max(1L, depth / min_chunk); ++c) {
        const long zc = (depth + c - 1) / c, wgs = tiles * ((depth + zc - 1) / zc);
        const double x = (double) wgs / (double) slots;
        const double whole = (double) (long) x, part = x - whole;
        const double rounds = 1.03 * whole + (part > 1e-9 ? 0.65 + 0.4 * part : 0.0);
        const double c_steps = rounds * (double) (zc + S);
        if (best == 0.0 || c_steps < best) {
            best = c_steps;
            best_c = c;
        }
    }
    if (cost) *cost = best;
    return (int) ((depth + best_c - 1) / best_c);
}

// Do spans pay?  By measurement (tools/spans_sweep.py, profiles/r04_spans_sweep_*.jsonl, r04_spans_vs_chunks_pmc.txt), not
// by the step counts alone: spans always take fewer CYCLES (768^3 bf16: 14.2 M against 15.8 M per launch), but neighbouring
// tiles are no longer at the same depth at the same time, so the rows they share and the 128-byte lines their misaligned
// row pieces straddle come from HBM twice (768^3 bf16: 1.82 GB fetched against 1.19 GB) -- and with that traffic the clock
// sinks (1.77 GHz from the third launch on against 2.0 GHz: 1000 us against 867 us).  In short bursts the fp64 separable
// box gained 5 - 18 % at 768^3 / 512^3; over the 50 sweeps of a bench.py run it did not (790 - 900 against a steady
// 851 - 865 GStencils/s at 768^3, tools/spans_bench_ab.sh) -- the same clock effect on a slower fuse.  So: spans where their
// busiest workgroup runs fewer steps than the chunked launch's (whole rounds of workgroups) AND the region's
// input is small enough for the 256 MB Infinity Cache to serve the second fetches (`bytes_cap`; bf16: + 5 - 12 % on
// 64 - 192 x 768^2, - 2.5 % at 384, - 20 % at 768; fp64 star: + 27 % on 32 x 512^2, + 6 % on 128 x 512^2, - 6 - 10 % at 512^3 and
// 768^3) -- the z-slabs and end regions of the multi-GPU drivers, mostly.
inline bool spans_pay(long tiles, long depth, int S, long slots, int zc_model, double bytes, double bytes_cap) {
    const long wgs = tiles * ((depth + zc_model - 1) / zc_model), rounds = (wgs + slots - 1) / slots;
    const double chunk_steps = (double) rounds * (double) (zc_model + S);
    const double span_steps = 1.5 * S + (double) (tiles * (depth + S)) / (double) slots;  // a start, the share, half a second start
    return span_steps < chunk_steps && (bytes_cap <= 0.0 || bytes <= bytes_cap);
}

// TEAM spans: the line runs over tile ROWS and is cut into one piece per team of tiles_x workgroups, one per tile of the
// row.  The tiles of a row stay at the same depth -- and, dealt out contiguously, in the same XCD's L2 -- so the lines their
// row pieces share are fetched once, as with chunks; what is given up is the rows shared with the tile rows above and
// below, and a team runs at the pace of its rim tiles.  Measured (tools/spans_sweep.py, tools/spans_bench_ab.sh): bf16
// 768^3 732 us per launch against 754 (chunks) and 954 (spans), 2115 - 2134 against 2055 - 2093 GStencils/s over 50 sweeps;
// fp64 star 512^3 and box 768^3 4 - 9 % SLOWER than chunks (their rows overlap by a quarter, the bf16 tile's by an eighth).
// So: the bf16 kernel, on regions too big for plain spans, where a team's busiest member runs at least 3 % fewer steps.
inline bool teams_pay(int tiles_x, int tiles_y, long depth, int S, long slots, int zc_model) {
    if (tiles_x > slots) return false;
    const long tiles = (long) tiles_x * tiles_y, wgs = tiles * ((depth + zc_model - 1) / zc_model), rounds = (wgs + slots - 1) / slots;
    const double chunk_steps = (double) rounds * (double) (zc_model + S);
    const double team_steps = 1.5 * S + (double) (tiles_y * (depth + S)) / (double) (slots / tiles_x);
    return team_steps < 0.97 * chunk_steps;
}

#ifdef __HIPCC__
#define LORA_SPANS_FN __host__ __device__ __forceinline__
#else
#define LORA_SPANS_FN inline
#endif
// (the decode below is what the kernels run; it is plain integer arithmetic and also compiles for the host, where
// lora_debug_span_cover replays it for the tests)
// tile t of the rim-first order -> (tx, ty): the TX x TY tiles' rim (first and last row, then first and last column), then
// the inner tiles row by row
LORA_SPANS_FN void rim_first_tile(int t, int TX, int TY, int &tx, int &ty) {
    if (TX < 3 || TY < 3) {
        ty = t / TX;
        tx = t - ty * TX;
        return;
    }
    const int rim = 2 * TX + 2 * (TY - 2);
    if (t < 2 * TX) {
        ty = t < TX ? 0 : TY - 1;
        tx = t < TX ? t : t - TX;
    } else if (t < rim) {
        const int k = t - 2 * TX;
        ty = 1 + (k >> 1);
        tx = (k & 1) ? TX - 1 : 0;
    } else {
        const int idx = t - rim;
        ty = 1 + idx / (TX - 2);
        tx = 1 + idx - (ty - 1) * (TX - 2);
    }
}

// workgroup `lin` of a chunked launch (`chunks` chunks per tile) -> (chunk, tile): all chunks of the rim tiles first
LORA_SPANS_FN void chunk_of(int lin, int chunks, int TX, int TY, int &chunk, int &tx, int &ty) {
    if (TX < 3 || TY < 3) {
        const int per_chunk = TX * TY;
        chunk = lin / per_chunk;
        rim_first_tile(lin - chunk * per_chunk, TX, TY, tx, ty);
        return;
    }
    const int rim = 2 * TX + 2 * (TY - 2), inner = (TX - 2) * (TY - 2);
    if (lin < rim * chunks) {
        chunk = lin / rim;
        rim_first_tile(lin - chunk * rim, TX, TY, tx, ty);
    } else {
        const int l2 = lin - rim * chunks;
        chunk = l2 / inner;
        rim_first_tile(rim + l2 - chunk * inner, TX, TY, tx, ty);
    }
}

// tile t of a tile column's line (spans_setup_column) -> tile row: the two rim rows first
LORA_SPANS_FN int column_line_row(int t, int TY) { return TY < 3 ? t : (t == 0 ? 0 : (t == 1 ? TY - 1 : t - 1)); }

// this workgroup's range [v0, v1) of the line
LORA_SPANS_FN void span_range(const Spans &sp, int lin, unsigned &v0, unsigned &v1) {
    v0 = (unsigned) lin * (unsigned) sp.q + (unsigned) (lin < sp.r ? lin : sp.r);
    v1 = v0 + (unsigned) sp.q + (lin < sp.r ? 1u : 0u);
}

// The next segment of [v0, v1): its tile and planes [z0, z0 + zc) of the region; advances v0.  A tile's first S steps'
// worth of units stand for the start every segment pays (charged once per workgroup outside the line, so a range that
// ENTERS a tile spends them here).  false: this piece of the range holds no plane (it ends inside a tile's start).
LORA_SPANS_FN bool span_next(const Spans &sp, int S, unsigned &v0, unsigned v1, int TX, int TY, int &tx, int &ty,
                                          int &z0, int &zc) {
    const unsigned rim_units = (unsigned) sp.nrim * (unsigned) sp.vt * (unsigned) sp.wrim;
    const bool in_rim = v0 < rim_units;
    const unsigned w = in_rim ? sp.wrim : sp.win, tile_units = (unsigned) sp.vt * w;
    const unsigned base = in_rim ? 0u : rim_units, ti = (v0 - base) / tile_units;
    const unsigned tile_v = base + ti * tile_units, e = v1 < tile_v + tile_units ? v1 : tile_v + tile_units;
    const int p0 = (int) ((v0 - tile_v) / w) - S, p1 = (int) ((e - tile_v) / w) - S;
    z0 = p0 > 0 ? p0 : 0;
    const int z1 = p1 > 0 ? p1 : 0;
    rim_first_tile((int) ti + (in_rim ? 0 : sp.nrim), TX, TY, tx, ty);
    v0 = e;
    zc = z1 - z0;
    return zc > 0;
}
}  // namespace lora
