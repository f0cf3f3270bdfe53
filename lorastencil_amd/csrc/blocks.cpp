// blocks.cpp -- two-axis block decomposition behind the C ABI (include/lorastencil.h, group E).
//
// The reference has no multi-GPU path at all; what is distributed is its time-step loop (2d/gpu.cu:544-546,
// 3d/gpu_star.cu:177-181).  slab.cpp cuts the grid along its outermost dimension only.  Here a Pa x Pb process grid cuts
// the TWO outer dimensions: rows x columns of the 2D shapes, planes x rows (z x y) of the 3D ones (the innermost dimension
// of a 3D grid stays whole: rows of it are what the kernels stream).  A block's local array is the ordinary padded layout
// with a ghost zone of G = radius x applications-per-launch x E cells on every side that has a neighbour (the pad IS the
// global halo on a side that is the global edge), so the block kernels are the single-GPU kernels on the local extents and
// ghost cells are plain interior cells of the local problem.  Why: a rank's share of the 8-GPU configurations runs 10 - 21 %
// faster on the device side as a 2 x 4 block than as a slab and refreshes 22 - 37 % fewer bytes
// (profiles/r04_block3d_shares.jsonl, r03_block_shares_loopback.jsonl).
//
//   * The refresh has two phases, so corners need no diagonal message: axis B first (the second dimension: G cells of the
//     OWN range of axis A -- strided, packed and unpacked by the block-copy kernel), then axis A over the full local
//     extent of axis B, ghost cells included (whole rows / planes: contiguous, sent in place).
//   * Both phases are posted on a communication stream behind the launch that exhausts the ghost zone.  The NEXT launch
//     waits for phase B (every row needs its B-side ghost cells), sweeps its deep interior along axis A, and only then
//     waits for phase A and sweeps the two end regions: the larger, contiguous messages hide behind compute (deferred wait,
//     as in slab.cpp).
//   * Messages go through the same four-entry callback table as the slabs (lora_slab_comm: RCCL in production, the
//     in-process loopback for blocks that share a device); peer = ia x Pb + ib.
//   * lora_block_run_many drives several blocks from one host thread (tests, one process with several devices).
//
// lorastencil_amd/blocks.py is the same decomposition for the 2D shapes written against torch.distributed; both are
// tested against the oracle.  Reference boundary only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "engine.h"

namespace {

constexpr int kRingInput = 0, kRingZero = 1;

// [begin, end) of part k of n cells in `parts` nearly equal parts, every boundary a multiple of `multiple`
// (blocks.py: _split with multiple = 2)
int split(int n, int parts, int k, int multiple, int *b, int *e) {
    const int base = (n / parts) / multiple * multiple;
    if (base < multiple) return LORA_EINVAL;
    *b = k * base;
    *e = k == parts - 1 ? n : (k + 1) * base;
    return LORA_OK;
}

}  // namespace

struct lora_block {
    lora_plan *plan = nullptr;
    int shape = 0, nd = 0, dtype = LORA_F64, device = 0;
    int gdims[3] = {0, 0, 0}, ldims[3] = {0, 0, 0};
    int grid[2] = {1, 1}, coords[2] = {0, 0};
    int a0 = 0, a1 = 0, b0 = 0, b1 = 0;          // own ranges (global interior indices) along axis A / B
    int ga_lo = 0, ga_hi = 0, gb_lo = 0, gb_hi = 0;  // ghost depth per side (0 at a global edge)
    int ghost = 0, apps = 1, need = 0, every = 1, radius = 0, gran = 1;
    int ha = 0, hb = 0;                          // pads of axis A / B
    bool fused = false, defer_wait = true;
    size_t esize = 8, a_stride = 0, b_stride = 0, bytes = 0;  // bytes between cells of axis A / axis B; one local buffer
    void *buf[2] = {nullptr, nullptr};
    void *pack[4] = {nullptr, nullptr, nullptr, nullptr};  // axis-B strips: send low, recv low, send high, recv high
    size_t pack_bytes = 0;
    int peer_a_lo = -1, peer_a_hi = -1, peer_b_lo = -1, peer_b_hi = -1;
    int steps_done = 0, cur = 0, valid = 0;
    int ringstate[2] = {kRingInput, kRingZero};
    bool pending_b = false, pending_a = false;
    hipStream_t cs = nullptr, ms = nullptr;
    hipEvent_t ev_ready = nullptr, ev_b = nullptr, ev_a = nullptr;
    lora_slab_comm comm{};
    bool have_comm = false;
    long launches = 0, exchanges = 0;
};

namespace {

#define BLK_HIP(expr)                          \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) {               \
            lora::set_last_error(#expr, e__);  \
            return LORA_EHIP;                  \
        }                                      \
    } while (0)

int own_a(const lora_block *s) { return s->a1 - s->a0; }
int own_b(const lora_block *s) { return s->b1 - s->b0; }

int sweep(lora_block *s, int napps, const void *src, void *dst, int b, int e) {
    if (e <= b) return LORA_OK;
    return lora_plan_stepn_region(s->plan, napps, src, dst, b, e, s->cs);
}
// two ranges in one call (one launch where the kernel family takes two: lora_plan_stepn_region2)
int sweep2(lora_block *s, int napps, const void *src, void *dst, int b0, int e0, int b1, int e1) {
    return lora_plan_stepn_region2(s->plan, napps, src, dst, b0, e0, b1, e1, s->cs);
}

int wait_b(lora_block *s) {
    if (s->pending_b) {
        BLK_HIP(hipStreamWaitEvent(s->cs, s->ev_b, 0));
        s->pending_b = false;
    }
    return LORA_OK;
}
int flush(lora_block *s) {
    if (int rc = wait_b(s)) return rc;
    if (s->pending_a) {
        BLK_HIP(hipStreamWaitEvent(s->cs, s->ev_a, 0));
        s->pending_a = false;
    }
    return LORA_OK;
}

int set_ring(lora_block *s, int b, int what, int src) {
    if (s->ringstate[b] == what) return LORA_OK;
    const int rc = lora_plan_halo(s->plan, s->buf[b], what == kRingZero ? nullptr : s->buf[src],
                                  what == kRingZero ? LORA_HALO_ZERO : LORA_HALO_COPY, s->cs);
    if (rc != LORA_OK) return rc;
    s->ringstate[b] = what;
    return LORA_OK;
}

// A strip of G cells of axis B over the own range of axis A, between the local array and a contiguous buffer.  In units
// of doubles: a strip row is G x b_stride bytes (2D: G doubles; 3D: G rows of the padded innermost extent -- a multiple of
// 8 elements, so also whole doubles on bf16 grids), one per cell of axis A.
int copy_strip(lora_block *s, void *t, int b_cell, void *contig, bool pack) {
    const long rows = own_a(s), cols = (long) (s->ghost * s->b_stride / 8), ld = (long) (s->a_stride / 8);
    double *arr = reinterpret_cast<double *>(static_cast<char *>(t) + (size_t) (s->ha + s->ga_lo) * s->a_stride + (size_t) b_cell * s->b_stride);
    double *c = static_cast<double *>(contig);
    const hipError_t e = pack ? lora::launch_copy_block(c, cols, arr, ld, rows, cols, s->ms) : lora::launch_copy_block(arr, ld, c, cols, rows, cols, s->ms);
    if (e != hipSuccess) {
        lora::set_last_error("block copy kernel launch", e);
        return LORA_EHIP;
    }
    return LORA_OK;
}

// the two phases of a refresh of buffer `t` of every block, posted on the communication streams
int exchange_all(lora_block **ss, int n, bool use_cur) {
    bool any = false;
    for (int i = 0; i < n; ++i)
        any = any || ss[i]->peer_a_lo >= 0 || ss[i]->peer_a_hi >= 0 || ss[i]->peer_b_lo >= 0 || ss[i]->peer_b_hi >= 0;
    if (!any) return LORA_OK;
    const lora_slab_comm &c0 = ss[0]->comm;
    for (int i = 0; i < n; ++i) {
        lora_block *s = ss[i];
        BLK_HIP(hipSetDevice(s->device));
        BLK_HIP(hipEventRecord(s->ev_ready, s->cs));
        BLK_HIP(hipStreamWaitEvent(s->ms, s->ev_ready, 0));
    }
    // ---- phase B: own cells next to the B-side cuts -> the neighbours' ghost cells there ----
    for (int i = 0; i < n; ++i) {
        lora_block *s = ss[i];
        BLK_HIP(hipSetDevice(s->device));
        void *t = s->buf[use_cur ? s->cur : 1 - s->cur];
        if (s->peer_b_lo >= 0)
            if (int rc = copy_strip(s, t, s->hb + s->gb_lo, s->pack[0], true)) return rc;
        if (s->peer_b_hi >= 0)
            if (int rc = copy_strip(s, t, s->hb + s->gb_lo + own_b(s) - s->ghost, s->pack[2], true)) return rc;
    }
    if (int rc = c0.group_begin(c0.ctx)) return rc;
    int rc = LORA_OK;
    for (int i = 0; i < n && rc == LORA_OK; ++i) {
        lora_block *s = ss[i];
        (void) hipSetDevice(s->device);
        const lora_slab_comm &c = s->comm;
        if (s->peer_b_lo >= 0 && rc == LORA_OK) rc = c.send(c.ctx, s->pack[0], s->pack_bytes, s->peer_b_lo, s->ms);
        if (s->peer_b_lo >= 0 && rc == LORA_OK) rc = c.recv(c.ctx, s->pack[1], s->pack_bytes, s->peer_b_lo, s->ms);
        if (s->peer_b_hi >= 0 && rc == LORA_OK) rc = c.send(c.ctx, s->pack[2], s->pack_bytes, s->peer_b_hi, s->ms);
        if (s->peer_b_hi >= 0 && rc == LORA_OK) rc = c.recv(c.ctx, s->pack[3], s->pack_bytes, s->peer_b_hi, s->ms);
    }
    int rc_end = c0.group_end(c0.ctx);
    if (rc != LORA_OK) return rc;
    if (rc_end != LORA_OK) return rc_end;
    for (int i = 0; i < n; ++i) {
        lora_block *s = ss[i];
        BLK_HIP(hipSetDevice(s->device));
        void *t = s->buf[use_cur ? s->cur : 1 - s->cur];
        if (s->peer_b_lo >= 0)
            if (int rc2 = copy_strip(s, t, s->hb, s->pack[1], false)) return rc2;
        if (s->peer_b_hi >= 0)
            if (int rc2 = copy_strip(s, t, s->hb + s->gb_lo + own_b(s), s->pack[3], false)) return rc2;
        BLK_HIP(hipEventRecord(s->ev_b, s->ms));
        s->pending_b = true;
    }
    // ---- phase A: whole rows / planes, B-side ghost cells included (what phase B just delivered travels on) ----
    if (int rc2 = c0.group_begin(c0.ctx)) return rc2;
    rc = LORA_OK;
    for (int i = 0; i < n && rc == LORA_OK; ++i) {
        lora_block *s = ss[i];
        (void) hipSetDevice(s->device);
        const lora_slab_comm &c = s->comm;
        char *base = static_cast<char *>(s->buf[use_cur ? s->cur : 1 - s->cur]);
        const size_t g = (size_t) s->ghost * s->a_stride;
        const size_t first = (size_t) (s->ha + s->ga_lo) * s->a_stride, last = first + (size_t) own_a(s) * s->a_stride;
        if (s->peer_a_lo >= 0 && rc == LORA_OK) rc = c.send(c.ctx, base + first, g, s->peer_a_lo, s->ms);
        if (s->peer_a_lo >= 0 && rc == LORA_OK) rc = c.recv(c.ctx, base + first - g, g, s->peer_a_lo, s->ms);
        if (s->peer_a_hi >= 0 && rc == LORA_OK) rc = c.send(c.ctx, base + last - g, g, s->peer_a_hi, s->ms);
        if (s->peer_a_hi >= 0 && rc == LORA_OK) rc = c.recv(c.ctx, base + last, g, s->peer_a_hi, s->ms);
    }
    rc_end = c0.group_end(c0.ctx);
    if (rc != LORA_OK) return rc;
    if (rc_end != LORA_OK) return rc_end;
    for (int i = 0; i < n; ++i) {
        lora_block *s = ss[i];
        BLK_HIP(hipSetDevice(s->device));
        BLK_HIP(hipEventRecord(s->ev_a, s->ms));
        s->pending_a = true;
        ++s->exchanges;
    }
    return LORA_OK;
}

// One launch of `napps` applications on every block (all blocks are at the same time level and ghost validity).
int launch_all(lora_block **ss, int n, int napps) {
    lora_block *s0 = ss[0];
    const bool fusedl = napps > 1;
    const int need = s0->radius * napps;
    for (int i = 0; i < n; ++i) {
        lora_block *s = ss[i];
        BLK_HIP(hipSetDevice(s->device));
        const int src = s->cur, dst = 1 - s->cur;
        // fused launches need the level-0 ring in both buffers; a single sweep from an even level writes the odd level,
        // whose ring is 0; from an odd level it writes an even one (ring = input)
        const bool even = s->steps_done % 2 == 0;
        const int want = (fusedl || !even) ? kRingInput : kRingZero;
        if (s->ringstate[dst] != want) {
            if (int rc = flush(s)) return rc;  // the ring copy reads the source's pads, next to ghost cells in flight
            if (int rc = set_ring(s, dst, want, src)) return rc;
        }
    }
    const bool cut = s0->grid[0] * s0->grid[1] > 1;
    if (cut && s0->valid < need) {
        lora::set_last_error_text("block driver: ghost zone exhausted");
        return LORA_EINVAL;
    }
    const int left = cut ? s0->valid - need : 0;
    const bool exchange = cut && left < s0->need;  // not enough for another full-depth launch: refresh behind this one
    for (int i = 0; i < n; ++i) {
        lora_block *s = ss[i];
        BLK_HIP(hipSetDevice(s->device));
        const void *src = s->buf[s->cur];
        void *dst = s->buf[1 - s->cur];
        // rows / planes of axis A that are still needed later: own cells + `left` ghost cells per cut side (the kernels
        // sweep the whole extent of axis B: its ghost cells beyond `left` cost nothing extra to compute and are never read)
        const int lo = s->ga_lo - (s->peer_a_lo >= 0 ? std::min(left, s->ga_lo) : 0);
        const int hi = s->ga_lo + own_a(s) + (s->peer_a_hi >= 0 ? std::min(left, s->ga_hi) : 0);
        // output cells whose inputs along axis A are own cells only: they need phase B of a refresh in flight, not phase A
        int a = s->ga_lo + need, b = s->ga_lo + own_a(s) - need;
        a = (a + s->gran - 1) / s->gran * s->gran;
        if (s->pending_a && s->defer_wait && b - a >= 2 * need && a >= lo && b <= hi) {
            if (int rc = wait_b(s)) return rc;
            if (int rc = sweep(s, napps, src, dst, a, b)) return rc;
            if (int rc = flush(s)) return rc;
            if (int rc = sweep2(s, napps, src, dst, lo / s->gran * s->gran, a, b, hi)) return rc;
        } else {
            if (int rc = flush(s)) return rc;
            if (int rc = sweep(s, napps, src, dst, lo / s->gran * s->gran, hi)) return rc;
        }
        s->valid = left;
    }
    if (exchange) {
        if (int rc = exchange_all(ss, n, false)) return rc;
        for (int i = 0; i < n; ++i) {
            if (!ss[i]->defer_wait) {
                (void) hipSetDevice(ss[i]->device);
                if (int rc = flush(ss[i])) return rc;
            }
            ss[i]->valid = ss[i]->ghost;
        }
    }
    for (int i = 0; i < n; ++i) {
        ss[i]->cur = 1 - ss[i]->cur;
        ss[i]->steps_done += napps;
        ++ss[i]->launches;
    }
    return LORA_OK;
}

}  // namespace

extern "C" {

void lora_block_destroy(lora_block *s) {
    if (!s) return;
    (void) hipSetDevice(s->device);
    if (s->cs) (void) hipStreamSynchronize(s->cs);
    if (s->ms) (void) hipStreamSynchronize(s->ms);
    for (void *b : s->buf)
        if (b) (void) hipFree(b);
    for (void *b : s->pack)
        if (b) (void) hipFree(b);
    for (hipEvent_t e : {s->ev_ready, s->ev_b, s->ev_a})
        if (e) (void) hipEventDestroy(e);
    if (s->cs) (void) hipStreamDestroy(s->cs);
    if (s->ms) (void) hipStreamDestroy(s->ms);
    lora_plan_destroy(s->plan);
    delete s;
}

int lora_block_create(lora_block **out, const lora_block_desc *d, const lora_slab_comm *comm) {
    if (!out || !d) return LORA_EINVAL;
    *out = nullptr;
    const int nd = lora_shape_ndim(d->shape);
    if (nd != 2 && nd != 3) {
        lora::set_last_error_text("block decomposition: 2D shapes (rows x columns) and 3D shapes (planes x rows)");
        return LORA_EUNSUPPORTED;
    }
    if (d->grid[0] < 1 || d->grid[1] < 1 || d->coords[0] < 0 || d->coords[0] >= d->grid[0] || d->coords[1] < 0 || d->coords[1] >= d->grid[1])
        return LORA_EINVAL;
    if (d->dtype == LORA_BF16 && nd != 3) return LORA_EUNSUPPORTED;
    const bool cut = d->grid[0] * d->grid[1] > 1;
    if (cut && !comm) return LORA_EINVAL;
    if (lora_device_count() <= 0) {
        lora::set_last_error_text("no HIP device visible");
        return LORA_ENODEVICE;
    }
    std::unique_ptr<lora_block, void (*)(lora_block *)> s(new (std::nothrow) lora_block(), lora_block_destroy);
    if (!s) return LORA_ENOMEM;
    s->shape = d->shape;
    s->nd = nd;
    s->dtype = d->dtype;
    s->device = d->device;
    s->esize = d->dtype == LORA_BF16 ? 2 : 8;
    for (int k = 0; k < nd; ++k) s->gdims[k] = d->global_dims[k];
    for (int k = 0; k < 2; ++k) {
        s->grid[k] = d->grid[k];
        s->coords[k] = d->coords[k];
    }
    s->ha = nd == 3 ? 1 : 4;
    s->hb = nd == 3 ? 2 : 4;
    s->radius = nd == 3 ? 1 : 3;
    s->defer_wait = !(d->flags & LORA_SLAB_NO_DEFER);
    if (comm) {
        s->comm = *comm;
        s->have_comm = true;
    }
    BLK_HIP(hipSetDevice(s->device));
    // column cuts of a 2D grid at even indices (16-byte rows); rows of a 2D grid at multiples of its region granularity
    // are not needed -- a block sweeps regions of its LOCAL array, whose first row is what it is
    const int mult_a = 1, mult_b = nd == 2 ? 2 : 1;
    if (split(s->gdims[0], s->grid[0], s->coords[0], mult_a, &s->a0, &s->a1) != LORA_OK ||
        split(s->gdims[1], s->grid[1], s->coords[1], mult_b, &s->b0, &s->b1) != LORA_OK) {
        lora::set_last_error_text("block decomposition: more parts than cells");
        return LORA_EINVAL;
    }
    int thinnest = 1 << 30;
    for (int k = 0; k < s->grid[0]; ++k) {
        int b, e;
        (void) split(s->gdims[0], s->grid[0], k, mult_a, &b, &e);
        if (s->grid[0] > 1) thinnest = std::min(thinnest, e - b);
    }
    for (int k = 0; k < s->grid[1]; ++k) {
        int b, e;
        (void) split(s->gdims[1], s->grid[1], k, mult_b, &b, &e);
        if (s->grid[1] > 1) thinnest = std::min(thinnest, e - b);
    }
    auto make_plan = [&](const int *dims, int spl, lora_plan **pl) -> int {
        int rc = lora_plan_create(pl, d->shape, d->dtype, dims, d->params);
        if (rc != LORA_OK) return rc;
        if (d->weights) rc = lora_plan_set_weights(*pl, d->weights, lora_shape_ntaps(d->shape));
        std::string opts = d->options ? d->options : "";
        size_t pos = 0;
        while (rc == LORA_OK && pos < opts.size()) {
            size_t comma = opts.find(',', pos);
            if (comma == std::string::npos) comma = opts.size();
            const std::string kv = opts.substr(pos, comma - pos);
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) return LORA_EINVAL;
            rc = lora_plan_set_option(*pl, kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1));
            pos = comma + 1;
        }
        if (rc == LORA_OK && spl > 0) rc = lora_plan_set_option(*pl, "steps_per_launch", spl);
        if (rc != LORA_OK) {
            lora_plan_destroy(*pl);
            *pl = nullptr;
        }
        return rc;
    };
    // applications per launch: what the kernels fuse on the GLOBAL grid (the same on every rank), reduced until a launch's
    // reach fits the thinnest block; then forced on the local plan
    int apps = 1;
    {
        lora_plan *probe = nullptr;
        int gd[3] = {s->gdims[0], s->gdims[1], nd > 2 ? s->gdims[2] : 0};
        if (int rc = make_plan(gd, 0, &probe)) return rc;
        (void) lora_plan_get_option(probe, "steps_per_launch", &apps);
        lora_plan_destroy(probe);
        if (d->flags & LORA_SLAB_NO_FUSION) apps = 1;
        if (nd == 3 && apps == 3) apps = 2;  // (block launches start at even steps on buffers that both carry the halo)
        while (apps > 1 && cut && thinnest < s->radius * apps) apps = apps >= 4 ? apps - 2 : 1;
    }
    s->apps = apps;
    s->fused = apps > 1;
    s->need = s->radius * apps;
    int every = d->exchange_every > 0 ? d->exchange_every : 2;
    if (cut) every = std::max(1, std::min(every, thinnest / s->need));
    if (cut && thinnest < s->need) {
        lora::set_last_error_text("blocks are thinner than the stencil radius");
        return LORA_EINVAL;
    }
    s->every = every;
    s->ghost = cut ? s->need * every : 0;
    s->ga_lo = s->coords[0] > 0 ? s->ghost : 0;
    s->ga_hi = s->coords[0] < s->grid[0] - 1 ? s->ghost : 0;
    s->gb_lo = s->coords[1] > 0 ? s->ghost : 0;
    s->gb_hi = s->coords[1] < s->grid[1] - 1 ? s->ghost : 0;
    const int pb = s->grid[1];
    s->peer_a_lo = s->ga_lo ? (s->coords[0] - 1) * pb + s->coords[1] : -1;
    s->peer_a_hi = s->ga_hi ? (s->coords[0] + 1) * pb + s->coords[1] : -1;
    s->peer_b_lo = s->gb_lo ? s->coords[0] * pb + s->coords[1] - 1 : -1;
    s->peer_b_hi = s->gb_hi ? s->coords[0] * pb + s->coords[1] + 1 : -1;
    s->ldims[0] = s->ga_lo + own_a(s.get()) + s->ga_hi;
    s->ldims[1] = s->gb_lo + own_b(s.get()) + s->gb_hi;
    s->ldims[2] = nd > 2 ? s->gdims[2] : 0;
    if (int rc = make_plan(s->ldims, apps, &s->plan)) return rc;
    int got = 0;
    (void) lora_plan_get_option(s->plan, "steps_per_launch", &got);
    if (got != apps) {
        lora::set_last_error_text("a block's plan cannot run the launch depth chosen for the grid");
        return LORA_EUNSUPPORTED;
    }
    s->gran = std::max(1, lora_plan_region_granularity(s->plan));
    s->bytes = lora_plan_padded_bytes(s->plan);
    s->a_stride = s->bytes / (size_t) (s->ldims[0] + 2 * s->ha);
    s->b_stride = nd == 2 ? s->esize : (size_t) (s->gdims[2] + 8) * s->esize;
    BLK_HIP(hipMalloc(&s->buf[0], s->bytes));
    BLK_HIP(hipMalloc(&s->buf[1], s->bytes));
    BLK_HIP(hipMemset(s->buf[0], 0, s->bytes));
    BLK_HIP(hipMemset(s->buf[1], 0, s->bytes));
    s->pack_bytes = (size_t) own_a(s.get()) * s->ghost * s->b_stride;
    if (s->pack_bytes % 8) return LORA_EUNSUPPORTED;
    for (int k = 0; k < 4; ++k)
        if ((k < 2 ? s->peer_b_lo : s->peer_b_hi) >= 0) BLK_HIP(hipMalloc(&s->pack[k], s->pack_bytes));
    BLK_HIP(hipStreamCreateWithFlags(&s->cs, hipStreamNonBlocking));
    BLK_HIP(hipStreamCreateWithFlags(&s->ms, hipStreamNonBlocking));
    BLK_HIP(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    BLK_HIP(hipEventCreateWithFlags(&s->ev_b, hipEventDisableTiming));
    BLK_HIP(hipEventCreateWithFlags(&s->ev_a, hipEventDisableTiming));
    s->valid = s->ghost;
    *out = s.release();
    return LORA_OK;
}

int lora_block_info(const lora_block *s, lora_block_info_t *info) {
    if (!s || !info) return LORA_EINVAL;
    std::memset(info, 0, sizeof *info);
    info->own_begin[0] = s->a0;
    info->own_end[0] = s->a1;
    info->own_begin[1] = s->b0;
    info->own_end[1] = s->b1;
    info->ghost = s->ghost;
    info->ghost_lo[0] = s->ga_lo;
    info->ghost_hi[0] = s->ga_hi;
    info->ghost_lo[1] = s->gb_lo;
    info->ghost_hi[1] = s->gb_hi;
    info->apps_per_launch = s->apps;
    info->exchange_every = s->every;
    info->steps_done = s->steps_done;
    for (int k = 0; k < s->nd; ++k) info->local_dims[k] = s->ldims[k];
    info->launches = s->launches;
    info->exchanges = s->exchanges;
    info->local_bytes = s->bytes;
    info->bytes_per_refresh = (size_t) ((s->peer_b_lo >= 0) + (s->peer_b_hi >= 0)) * s->pack_bytes +
                              (size_t) ((s->peer_a_lo >= 0) + (s->peer_a_hi >= 0)) * s->ghost * s->a_stride;
    return LORA_OK;
}

void *lora_block_buffer(lora_block *s, int which) {
    if (!s) return nullptr;
    return which == 0 || which == 1 ? s->buf[which] : s->buf[s->cur];
}
void *lora_block_stream(lora_block *s) { return s ? s->cs : nullptr; }
lora_plan *lora_block_plan(lora_block *s) { return s ? s->plan : nullptr; }

// buffer 0 <- this block's cells of the padded GLOBAL host array (ghost cells and pads included), buffer 1 <- 0
int lora_block_load(lora_block *s, const void *host_global_padded) {
    if (!s || !host_global_padded) return LORA_EINVAL;
    BLK_HIP(hipSetDevice(s->device));
    BLK_HIP(hipStreamSynchronize(s->cs));
    BLK_HIP(hipStreamSynchronize(s->ms));
    // local padded cell (pa, pb) = global padded cell (pa + a0 - ga_lo, pb + b0 - gb_lo); a local B-run is contiguous
    const size_t gb_stride = s->b_stride;  // bytes per cell of axis B: the same in both arrays
    const size_t g_a_stride = (size_t) (s->gdims[1] + 2 * s->hb) * gb_stride;
    const size_t run = (size_t) (s->ldims[1] + 2 * s->hb) * gb_stride;
    const char *g = static_cast<const char *>(host_global_padded) + (size_t) (s->a0 - s->ga_lo) * g_a_stride + (size_t) (s->b0 - s->gb_lo) * gb_stride;
    BLK_HIP(hipMemcpy2D(s->buf[0], run, g, g_a_stride, run, (size_t) (s->ldims[0] + 2 * s->ha), hipMemcpyHostToDevice));
    BLK_HIP(hipMemset(s->buf[1], 0, s->bytes));
    s->steps_done = 0;
    s->cur = 0;
    s->valid = s->ghost;
    s->ringstate[0] = kRingInput;
    s->ringstate[1] = kRingZero;
    s->pending_a = s->pending_b = false;
    return LORA_OK;
}

int lora_block_run_many(lora_block **ss, int n, int times) {
    if (!ss || n < 1 || times < 0) return LORA_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (!ss[i]) return LORA_EINVAL;
        if (ss[i]->apps != ss[0]->apps || ss[i]->ghost != ss[0]->ghost || ss[i]->steps_done != ss[0]->steps_done) return LORA_EINVAL;
        if (ss[i]->grid[0] * ss[i]->grid[1] > 1 && !ss[i]->have_comm) return LORA_EINVAL;
    }
    lora_block *s0 = ss[0];
    int t = 0;
    while (t < times) {
        const bool even = s0->steps_done % 2 == 0;
        int napps = 1;
        if (s0->fused && even && times - t >= s0->apps)
            napps = s0->apps;
        else if (s0->fused && even && s0->apps >= 4 && times - t >= 2)
            napps = (s0->nd == 2 && s0->apps == 6 && times - t >= 4) ? 4 : 2;  // (the slab driver's rule)
        if (int rc = launch_all(ss, n, napps)) return rc;
        t += napps;
    }
    return LORA_OK;
}
int lora_block_run(lora_block *s, int times) { return lora_block_run_many(&s, 1, times); }

int lora_block_sync(lora_block *s) {
    if (!s) return LORA_EINVAL;
    BLK_HIP(hipSetDevice(s->device));
    if (int rc = flush(s)) return rc;
    BLK_HIP(hipStreamSynchronize(s->cs));
    BLK_HIP(hipStreamSynchronize(s->ms));
    return LORA_OK;
}

// own cells of the current buffer -> the padded GLOBAL host array; a block at a global edge also returns its pads there
int lora_block_store(lora_block *s, void *host_global_padded) {
    if (!s || !host_global_padded) return LORA_EINVAL;
    if (int rc = lora_block_sync(s)) return rc;
    const size_t cb = s->b_stride;
    const size_t g_a_stride = (size_t) (s->gdims[1] + 2 * s->hb) * cb;
    const size_t l_a_stride = s->a_stride;
    const int la0 = s->ha + s->ga_lo - (s->peer_a_lo < 0 ? s->ha : 0), la1 = s->ha + s->ga_lo + own_a(s) + (s->peer_a_hi < 0 ? s->ha : 0);
    const int lb0 = s->hb + s->gb_lo - (s->peer_b_lo < 0 ? s->hb : 0), lb1 = s->hb + s->gb_lo + own_b(s) + (s->peer_b_hi < 0 ? s->hb : 0);
    char *g = static_cast<char *>(host_global_padded) + (size_t) (la0 + s->a0 - s->ga_lo) * g_a_stride + (size_t) (lb0 + s->b0 - s->gb_lo) * cb;
    const char *l = static_cast<const char *>(s->buf[s->cur]) + (size_t) la0 * l_a_stride + (size_t) lb0 * cb;
    BLK_HIP(hipMemcpy2D(g, g_a_stride, l, l_a_stride, (size_t) (lb1 - lb0) * cb, (size_t) (la1 - la0), hipMemcpyDeviceToHost));
    return LORA_OK;
}

}  // extern "C"
