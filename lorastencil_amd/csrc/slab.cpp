// slab.cpp -- multi-GPU time-step driver behind the C ABI (include/lorastencil.h, group D).
//
// The reference has no multi-GPU path (no nccl / mpi / streams anywhere); its time-step loop
// (2d/gpu.cu:544-546, 3d/gpu_star.cu:177-181) is what gets distributed here.  The grid is cut into slabs of the
// outermost interior dimension, one lora_slab per GPU.  A slab's local array is
//     [pad][G ghost rows][own rows][G ghost rows][pad]
// in the reference's padded layout, so the slab kernels are the single-GPU kernels (lora_plan_step*_region): ghost rows
// are ordinary interior rows of the local problem; a slab at a global edge has no ghost rows there and its pad keeps
// the reference's halo semantics.
//
//   * Ghost zones are G = radius x applications-per-launch x E rows deep and are refreshed from the neighbours' own rows
//     every E launches (E x fewer, E x larger messages; between refreshes a launch also sweeps the ghost rows that are
//     still needed later).
//   * On the launch that exhausts the ghost zone the two boundary strips are swept first, the exchange is posted on a
//     communication stream (ncclSend / ncclRecv inside one ncclGroupStart / End: RCCL over xGMI) and the interior is
//     swept meanwhile; the NEXT launch sweeps its deep interior before it waits for the ghost rows (deferred wait).
//   * The exchange goes through a small table of callbacks (lora_slab_comm): RCCL in production
//     (lora_slab_comm_rccl, librccl loaded on first use), an in-process loopback between slabs that share a device for
//     tests on a one-GPU box.
//   * lora_slab_run_many drives several slabs from ONE host thread (the CLIs' --gpus N: one process, N devices,
//     ncclCommInitAll); with one slab per process it is the usual one-process-per-GPU driver.
//
// lorastencil_amd/slab.py is the same schedule written against torch.distributed (bench.py, the gloo tests); the two
// are tested against each other.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "engine.h"

namespace {

constexpr int kRingInput = 0, kRingZero = 1;

int halo0_of(int nd) { return nd == 3 ? 1 : 4; }
int radius_of(int nd) { return nd == 1 ? 4 : (nd == 2 ? 3 : 1); }

// balanced contiguous split, every boundary a multiple of `multiple` (slab.py: slab_layout)
void split(int n0, int nranks, int rank, int multiple, int *begin, int *end) {
    const int units = (n0 + multiple - 1) / multiple;
    const int base = units / nranks, extra = units % nranks;
    const int su = rank * base + std::min(rank, extra);
    const int eu = su + base + (rank < extra ? 1 : 0);
    *begin = su * multiple;
    *end = std::min(eu * multiple, n0);
}

}  // namespace

struct lora_slab {
    lora_plan *plan = nullptr;
    int shape = 0, nd = 0, dtype = LORA_F64, device = 0;
    int gdims[3] = {0, 0, 0}, ldims[3] = {0, 0, 0};
    int rank = 0, nranks = 1;
    bool ring = false;  // a single slab closed into a ring (its own neighbour): rehearsal of the exchange path
    int begin = 0, end = 0, own = 0, ghost = 0, gt = 0, gb = 0, h0 = 0, radius = 0;
    int apps = 1, need = 0, every = 1, strip = 0;
    bool fused = false, dirichlet = false, overlap = true, defer_wait = true;
    int up = -1, down = -1;
    size_t esize = 8, row_bytes = 0, bytes = 0;
    void *buf[2] = {nullptr, nullptr};
    int steps_done = 0, cur = 0, valid = 0;
    int ringstate[2] = {kRingInput, kRingZero};
    bool pending = false;
    hipStream_t cs = nullptr, ms = nullptr;
    hipEvent_t ev_ready = nullptr, ev_exch = nullptr;
    lora_slab_comm comm{};
    bool have_comm = false;
    long launches = 0, exchanges = 0;
};

namespace {

#define SLAB_HIP(expr)                         \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) {               \
            lora::set_last_error(#expr, e__);  \
            return LORA_EHIP;                  \
        }                                      \
    } while (0)

int sweep(lora_slab *s, int napps, const void *src, void *dst, int b, int e) {
    if (e <= b) return LORA_OK;
    return lora_plan_stepn_region(s->plan, napps, src, dst, b, e, s->cs);
}
// two ranges in one call (one launch where the kernel family takes two: lora_plan_stepn_region2)
int sweep2(lora_slab *s, int napps, const void *src, void *dst, int b0, int e0, int b1, int e1) {
    return lora_plan_stepn_region2(s->plan, napps, src, dst, b0, e0, b1, e1, s->cs);
}

int flush(lora_slab *s) {
    if (s->pending) {
        SLAB_HIP(hipStreamWaitEvent(s->cs, s->ev_exch, 0));
        s->pending = false;
    }
    return LORA_OK;
}

int set_ring(lora_slab *s, int b, int what, int src) {
    if (s->ringstate[b] == what) return LORA_OK;
    const int rc = lora_plan_halo(s->plan, s->buf[b], what == kRingZero ? nullptr : s->buf[src],
                                  what == kRingZero ? LORA_HALO_ZERO : LORA_HALO_COPY, s->cs);
    if (rc != LORA_OK) return rc;
    s->ringstate[b] = what;
    return LORA_OK;
}

// own boundary rows of buffer t -> the neighbours' ghost zones; enqueued on the communication stream after the strips
int post_exchange(lora_slab *s, void *t) {
    char *base = static_cast<char *>(t);
    const size_t g = (size_t) s->ghost * s->row_bytes;
    const size_t first = (size_t) (s->h0 + s->gt) * s->row_bytes;
    const size_t last = first + (size_t) s->own * s->row_bytes;
    const lora_slab_comm &c = s->comm;
    int rc = LORA_OK;
    // order: with one peer on both sides (a ring of one) the k-th send to it meets its k-th receive from us -- our top
    // strip must land in ITS bottom ghost zone
    if (s->up >= 0 && rc == LORA_OK) rc = c.send(c.ctx, base + first, g, s->up, s->ms);
    if (s->down >= 0 && rc == LORA_OK) rc = c.recv(c.ctx, base + last, g, s->down, s->ms);
    if (s->down >= 0 && rc == LORA_OK) rc = c.send(c.ctx, base + last - g, g, s->down, s->ms);
    if (s->up >= 0 && rc == LORA_OK) rc = c.recv(c.ctx, base + first - g, g, s->up, s->ms);
    return rc;
}

int exchange_all(lora_slab **ss, int n, bool use_cur) {
    bool any = false;
    for (int i = 0; i < n; ++i) any = any || ss[i]->up >= 0 || ss[i]->down >= 0;
    if (!any) return LORA_OK;
    for (int i = 0; i < n; ++i) {
        lora_slab *s = ss[i];
        SLAB_HIP(hipSetDevice(s->device));
        SLAB_HIP(hipEventRecord(s->ev_ready, s->cs));
        SLAB_HIP(hipStreamWaitEvent(s->ms, s->ev_ready, 0));
    }
    const lora_slab_comm &c0 = ss[0]->comm;
    if (int rc = c0.group_begin(c0.ctx)) return rc;
    int rc = LORA_OK;
    for (int i = 0; i < n && rc == LORA_OK; ++i) {
        lora_slab *s = ss[i];
        (void) hipSetDevice(s->device);
        rc = post_exchange(s, s->buf[use_cur ? s->cur : 1 - s->cur]);
    }
    const int rc_end = c0.group_end(c0.ctx);
    if (rc != LORA_OK) return rc;
    if (rc_end != LORA_OK) return rc_end;
    for (int i = 0; i < n; ++i) {
        lora_slab *s = ss[i];
        SLAB_HIP(hipSetDevice(s->device));
        SLAB_HIP(hipEventRecord(s->ev_exch, s->ms));
        s->pending = true;
        ++s->exchanges;
    }
    return LORA_OK;
}

// One launch of `napps` applications on every slab (all slabs are at the same time level and ghost validity).
int launch_all(lora_slab **ss, int n, int napps) {
    lora_slab *s0 = ss[0];
    const bool fusedl = napps > 1;
    const int need = s0->radius * napps;
    for (int i = 0; i < n; ++i) {
        lora_slab *s = ss[i];
        SLAB_HIP(hipSetDevice(s->device));
        const int src = s->cur, dst = 1 - s->cur;
        int want = -1;
        if (s->dirichlet)
            want = kRingInput;  // fixed boundary: every level carries the caller's halo ring
        else if (s->fused || fusedl) {
            // fused launches need the level-0 ring in both buffers; a single sweep from an even level writes the odd
            // level, whose ring is 0; from an odd level it writes an even one (ring = input)
            const bool even = s->steps_done % 2 == 0;
            want = (fusedl || !even) ? kRingInput : kRingZero;
        }
        if (want >= 0 && s->ringstate[dst] != want) {
            if (int rc = flush(s)) return rc;  // the ring copy reads the source's halo columns, ghost rows included
            if (int rc = set_ring(s, dst, want, src)) return rc;
        }
    }
    const bool split = s0->up >= 0 || s0->down >= 0 || n > 1;
    if (!split) {
        for (int i = 0; i < n; ++i) {
            lora_slab *s = ss[i];
            SLAB_HIP(hipSetDevice(s->device));
            if (int rc = sweep(s, napps, s->buf[s->cur], s->buf[1 - s->cur], 0, s->own)) return rc;
        }
    } else {
        if (s0->valid < need) {
            lora::set_last_error_text("slab driver: ghost zone exhausted");
            return LORA_EINVAL;
        }
        const int left = s0->valid - need;
        const bool exchange = left < s0->need;  // not enough for another launch of this driver's kind: refresh now
        if (exchange) {
            bool overlapped = false;
            for (int i = 0; i < n; ++i) {
                lora_slab *s = ss[i];
                SLAB_HIP(hipSetDevice(s->device));
                if (int rc = flush(s)) return rc;
                const void *src = s->buf[s->cur];
                void *dst = s->buf[1 - s->cur];
                const int st = s->strip;
                if (s->overlap && s->own > 2 * st) {
                    overlapped = true;
                    if (int rc = sweep2(s, napps, src, dst, s->gt, s->gt + (s->up >= 0 ? st : 0), s->gt + s->own - (s->down >= 0 ? st : 0),
                                        s->gt + s->own))
                        return rc;
                } else {
                    if (int rc = sweep(s, napps, src, dst, s->gt, s->gt + s->own)) return rc;
                }
            }
            if (int rc = exchange_all(ss, n, false)) return rc;
            for (int i = 0; i < n && overlapped; ++i) {
                lora_slab *s = ss[i];
                SLAB_HIP(hipSetDevice(s->device));
                const int st = s->strip;
                if (s->overlap && s->own > 2 * st)
                    if (int rc = sweep(s, napps, s->buf[s->cur], s->buf[1 - s->cur], s->gt + (s->up >= 0 ? st : 0),
                                       s->gt + s->own - (s->down >= 0 ? st : 0)))
                        return rc;
            }
            // The ghost rows in flight are first read by the NEXT launch, and only within `need` rows of the slab's
            // ends: they stay pending -- that launch sweeps its deep interior before it waits for them.
            for (int i = 0; i < n; ++i) {
                if (!ss[i]->defer_wait) {
                    (void) hipSetDevice(ss[i]->device);
                    if (int rc = flush(ss[i])) return rc;
                }
                ss[i]->valid = ss[i]->ghost;
            }
        } else {
            for (int i = 0; i < n; ++i) {
                lora_slab *s = ss[i];
                SLAB_HIP(hipSetDevice(s->device));
                const void *src = s->buf[s->cur];
                void *dst = s->buf[1 - s->cur];
                const int lo = s->gt - (s->up >= 0 ? left : 0), hi = s->gt + s->own + (s->down >= 0 ? left : 0);
                int a = s->gt + need, b = s->gt + s->own - need;  // output rows whose inputs are own rows only
                if (s->nd == 1) {  // 1D regions start on even points
                    a += a & 1;
                    b -= b & 1;
                }
                if (s->pending && b - a >= 2 * need) {
                    if (int rc = sweep(s, napps, src, dst, a, b)) return rc;
                    if (int rc = flush(s)) return rc;
                    if (int rc = sweep2(s, napps, src, dst, lo, a, b, hi)) return rc;
                } else {
                    if (int rc = flush(s)) return rc;
                    if (int rc = sweep(s, napps, src, dst, lo, hi)) return rc;
                }
                s->valid = left;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        ss[i]->cur = 1 - ss[i]->cur;
        ss[i]->steps_done += napps;
        ++ss[i]->launches;
    }
    return LORA_OK;
}

// ---- RCCL backend (librccl loaded on first use: the engine library does not link it) ---------------------------
struct Rccl {
    void *lib = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.GroupStart = reinterpret_cast<int (*)()>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<int (*)()>(dlsym(r.lib, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.CommInitAll && r.CommDestroy;
    });
    return r;
}

int rccl_fail(const char *what, int code) {
    Rccl &r = rccl();
    std::string t = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error");
    lora::set_last_error_text(t.c_str());
    return LORA_EHIP;
}
int rccl_group_begin(void *) {
    const int rc = rccl().GroupStart();
    return rc ? rccl_fail("ncclGroupStart", rc) : LORA_OK;
}
int rccl_group_end(void *) {
    const int rc = rccl().GroupEnd();
    return rc ? rccl_fail("ncclGroupEnd", rc) : LORA_OK;
}
int rccl_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream) {
    const int rc = rccl().Send(buf, bytes, /* ncclInt8 */ 0, peer, ctx, static_cast<hipStream_t>(stream));
    return rc ? rccl_fail("ncclSend", rc) : LORA_OK;
}
int rccl_recv(void *ctx, void *buf, size_t bytes, int peer, void *stream) {
    const int rc = rccl().Recv(buf, bytes, /* ncclInt8 */ 0, peer, ctx, static_cast<hipStream_t>(stream));
    return rc ? rccl_fail("ncclRecv", rc) : LORA_OK;
}

// ---- loopback backend: slabs of ONE process that share a device exchange by device-to-device copies -------------
struct LoopShared {
    struct Msg {
        int src, dst;
        const void *buf;
        size_t bytes;
        hipEvent_t ready;
        hipStream_t stream;  // the sender's communication stream: made to wait for the copy (its "send complete")
    };
    struct Want {
        int src, dst;
        void *buf;
        size_t bytes;
        hipStream_t stream;
    };
    std::vector<Msg> sends;
    std::vector<Want> recvs;
    int depth = 0;
};
struct LoopCtx {
    std::shared_ptr<LoopShared> sh;
    int rank;
};

int loop_group_begin(void *ctx) {
    ++static_cast<LoopCtx *>(ctx)->sh->depth;
    return LORA_OK;
}
int loop_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream) {
    LoopCtx *c = static_cast<LoopCtx *>(ctx);
    hipEvent_t ev = nullptr;
    SLAB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    SLAB_HIP(hipEventRecord(ev, static_cast<hipStream_t>(stream)));  // the sender's strips are complete here
    c->sh->sends.push_back({c->rank, peer, buf, bytes, ev, static_cast<hipStream_t>(stream)});
    return LORA_OK;
}
int loop_recv(void *ctx, void *buf, size_t bytes, int peer, void *stream) {
    LoopCtx *c = static_cast<LoopCtx *>(ctx);
    c->sh->recvs.push_back({peer, c->rank, buf, bytes, static_cast<hipStream_t>(stream)});
    return LORA_OK;
}
int loop_group_end(void *ctx) {
    LoopShared &sh = *static_cast<LoopCtx *>(ctx)->sh;
    if (--sh.depth > 0) return LORA_OK;
    int status = LORA_OK;
    // the k-th receive of (src -> dst) takes the k-th send of (src -> dst), like NCCL
    for (LoopShared::Want &w : sh.recvs) {
        bool found = false;
        for (LoopShared::Msg &m : sh.sends) {
            if (m.buf && m.src == w.src && m.dst == w.dst) {
                if (m.bytes != w.bytes) status = LORA_EINVAL;
                hipEvent_t done = nullptr;
                if (hipStreamWaitEvent(w.stream, m.ready, 0) != hipSuccess ||
                    hipMemcpyAsync(w.buf, m.buf, w.bytes, hipMemcpyDeviceToDevice, w.stream) != hipSuccess ||
                    hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess ||
                    hipEventRecord(done, w.stream) != hipSuccess || hipStreamWaitEvent(m.stream, done, 0) != hipSuccess)
                    status = LORA_EHIP;
                if (done) (void) hipEventDestroy(done);
                m.buf = nullptr;
                found = true;
                break;
            }
        }
        if (!found) status = LORA_EINVAL;
    }
    for (LoopShared::Msg &m : sh.sends) {
        if (m.buf) status = LORA_EINVAL;  // a send nobody received
        (void) hipEventDestroy(m.ready);  // destruction is deferred by the runtime until the waits have passed
    }
    sh.sends.clear();
    sh.recvs.clear();
    if (status != LORA_OK) lora::set_last_error_text("loopback exchange: unmatched or failed message");
    return status;
}

std::vector<std::unique_ptr<LoopCtx>> g_loop_ctxs;  // live for the process (tests / the CLI's loopback rehearsal)
std::mutex g_loop_mutex;

}  // namespace

extern "C" {

int lora_slab_comm_rccl(lora_slab_comm *out, void *nccl_comm) {
    if (!out || !nccl_comm) return LORA_EINVAL;
    if (!rccl().ok) {
        lora::set_last_error_text("librccl.so could not be loaded");
        return LORA_EUNSUPPORTED;
    }
    out->ctx = nccl_comm;
    out->group_begin = rccl_group_begin;
    out->send = rccl_send;
    out->recv = rccl_recv;
    out->group_end = rccl_group_end;
    return LORA_OK;
}

int lora_slab_comm_loopback(lora_slab_comm *out, int nranks) {
    if (!out || nranks < 1) return LORA_EINVAL;
    auto sh = std::make_shared<LoopShared>();
    std::lock_guard<std::mutex> lock(g_loop_mutex);
    for (int r = 0; r < nranks; ++r) {
        g_loop_ctxs.emplace_back(new LoopCtx{sh, r});
        out[r].ctx = g_loop_ctxs.back().get();
        out[r].group_begin = loop_group_begin;
        out[r].send = loop_send;
        out[r].recv = loop_recv;
        out[r].group_end = loop_group_end;
    }
    return LORA_OK;
}

void lora_slab_destroy(lora_slab *s) {
    if (!s) return;
    (void) hipSetDevice(s->device);
    if (s->cs) (void) hipStreamSynchronize(s->cs);
    if (s->ms) (void) hipStreamSynchronize(s->ms);
    for (void *b : s->buf)
        if (b) (void) hipFree(b);
    if (s->ev_ready) (void) hipEventDestroy(s->ev_ready);
    if (s->ev_exch) (void) hipEventDestroy(s->ev_exch);
    if (s->cs) (void) hipStreamDestroy(s->cs);
    if (s->ms) (void) hipStreamDestroy(s->ms);
    lora_plan_destroy(s->plan);
    delete s;
}

int lora_slab_create(lora_slab **out, const lora_slab_desc *d, const lora_slab_comm *comm) {
    if (!out || !d) return LORA_EINVAL;
    *out = nullptr;
    const int nd = lora_shape_ndim(d->shape);
    if (nd == 0 || d->nranks < 1 || d->rank < 0 || d->rank >= d->nranks) return LORA_EINVAL;
    const bool ring = (d->flags & LORA_SLAB_RING_OF_ONE) != 0;
    if (ring && d->nranks != 1) return LORA_EINVAL;
    if (d->boundary != LORA_BC_REFERENCE && d->boundary != LORA_BC_DIRICHLET) {
        lora::set_last_error_text("the C++ slab driver takes the reference and the Dirichlet boundary");
        return LORA_EUNSUPPORTED;
    }
    const bool splitting = d->nranks > 1 || ring;
    if (splitting && !comm) return LORA_EINVAL;
    if (lora_device_count() <= 0) {
        lora::set_last_error_text("no HIP device visible");
        return LORA_ENODEVICE;
    }
    std::unique_ptr<lora_slab, void (*)(lora_slab *)> s(new (std::nothrow) lora_slab(), lora_slab_destroy);
    if (!s) return LORA_ENOMEM;
    s->shape = d->shape;
    s->nd = nd;
    s->dtype = d->dtype;
    s->rank = d->rank;
    s->nranks = d->nranks;
    s->ring = ring;
    s->device = d->device;
    for (int k = 0; k < nd; ++k) s->gdims[k] = d->global_dims[k];
    s->h0 = halo0_of(nd);
    s->radius = radius_of(nd);
    s->dirichlet = d->boundary == LORA_BC_DIRICHLET;
    // Boundary strips first is the default in 1D only.  In 2D and 3D the whole slab goes in one launch unless
    // LORA_SLAB_OVERLAP asks for the strips: the fused kernels run ONE round of workgroups sized to their region, and a
    // chunk pays its warm-up steps (41 in 2D, 9 planes in 3D) whatever its length, so two 32-row strips cost two
    // third-length launches on a nearly empty chip
    // (ring of one over RCCL, GStencils/s per rank, strips + deferred wait / whole slab + deferred wait / whole slab +
    // wait at once: 2048-row share 789 / 879 / 930, 4096 rows 973 / 1114 / 1131, 8192 rows 1234 / 1251 / 1288 at a
    // refresh every 4 launches -- tools/cslab_overlap.py, profiles/r03_cslab_schedule_sweep.jsonl).  The deferred wait
    // stays: it is what hides a real link behind the next launch's interior.
    // (3D, same sweep: 64-plane star share 329-424 with the strips, 436-521 whole; 96-plane box share 627-640 / 661-687)
    s->overlap = nd >= 2 ? (d->flags & LORA_SLAB_OVERLAP) != 0 : !(d->flags & LORA_SLAB_NO_OVERLAP);
    s->defer_wait = !(d->flags & LORA_SLAB_NO_DEFER);
    if (comm) {
        s->comm = *comm;
        s->have_comm = true;
    }
    SLAB_HIP(hipSetDevice(s->device));
    const int multiple = nd == 1 ? 2 : (nd == 2 ? 32 : 1);
    int thinnest = d->global_dims[0];
    for (int r = 0; r < d->nranks; ++r) {
        int b, e;
        split(d->global_dims[0], d->nranks, r, multiple, &b, &e);
        thinnest = std::min(thinnest, e - b);
    }
    if ((d->global_dims[0] + multiple - 1) / multiple < d->nranks) {
        lora::set_last_error_text("cannot split the outermost extent into that many slabs");
        return LORA_EINVAL;
    }
    split(d->global_dims[0], d->nranks, d->rank, multiple, &s->begin, &s->end);
    s->own = s->end - s->begin;

    // Applications per launch: what the kernels of this shape fuse (8 in 1D, 4 / 2 in 2D, 2 in 3D), reduced until a
    // launch's reach fits the thinnest slab; the ghost depth follows from it.  A probe plan on the global dims answers.
    auto make_plan = [&](const int *dims, int spl, lora_plan **pl) -> int {
        const int old_bc = lora_set_default_boundary(d->boundary);
        int rc = lora_plan_create(pl, d->shape, d->dtype, dims, d->params);
        (void) lora_set_default_boundary(old_bc);
        if (rc != LORA_OK) return rc;
        if (d->weights) rc = lora_plan_set_weights(*pl, d->weights, lora_shape_ntaps(d->shape));
        // "key=value,key=value": kernel options applied BEFORE the ghost depth is fixed
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
    int apps = 1;
    {
        lora_plan *probe = nullptr;
        int gd[3] = {d->global_dims[0], nd > 1 ? d->global_dims[1] : 0, nd > 2 ? d->global_dims[2] : 0};
        if (int rc = make_plan(gd, 0, &probe)) return rc;
        (void) lora_plan_get_option(probe, "steps_per_launch", &apps);
        lora_plan_destroy(probe);
        if (d->flags & LORA_SLAB_NO_FUSION) apps = 1;
        // slab launches start at even steps on buffers that both carry the halo: an even number of applications -- four
        // (register-resident kernel, big fp64 grids) or two; the three of the single-GPU plane-streaming schedule become two
        if (nd == 3 && apps == 3) apps = 2;
        // (2D slabs keep the plan's six per launch: ring-of-one shares of star2d1r 16384^2 at 8 GPUs, GStencils/s per rank,
        // six (workgroup-row kernel) / four (the same) / four (row-streaming): 847 / 825 / 703 -- tools/slab_shares.py)
        // 3D slabs keep the plan's four per launch whatever their thickness: second form of the register-resident kernel +
        // its chunk-length model + whole-slab launches, ring of one over RCCL, per rank, four / two per launch: 64-plane
        // share of star3d1r 512^3 479-521 / 357-380, 128 planes 635-650 / 430-470; 96-plane share of box3d1r 768^3
        // 661-687 / 434-448 (tools/cslab_3d_k.py, profiles/r03_cslab_3d_k.jsonl; the first form lost below 96 planes)
        while (apps > 1 && splitting && thinnest < s->radius * apps) apps = (nd >= 2 && apps >= 4) ? apps - 2 : 1;
    }
    s->apps = apps;
    s->fused = apps > 1;
    s->need = s->radius * apps;
    int every = d->exchange_every;
    if (every <= 0) {
        // refresh as rarely as keeps the redundant ghost sweeps within ~10 % of the thinnest slab.  (2D, six sweeps per
        // launch: a 2048-row share at 8 launches carries 14 % of ghost rows and still runs level with 4 launches / 7 % --
        // 891 vs 888 GStencils/s here, 964 vs 919 in slab.py, whose exchanges cost more host time -- because every
        // refresh also splits the launch after it in three for the deferred wait.)
        every = 8;
        while (every > 1 && (every - 1) * s->need > 0.1 * thinnest) every /= 2;
    }
    if (splitting) every = std::max(1, std::min(every, thinnest / s->need));
    if (splitting && thinnest < s->need) {
        lora::set_last_error_text("slabs are thinner than the stencil radius");
        return LORA_EINVAL;
    }
    s->every = every;
    s->ghost = splitting ? s->need * every : 0;
    s->gt = (s->rank > 0 || ring) ? s->ghost : 0;
    s->gb = (s->rank < s->nranks - 1 || ring) ? s->ghost : 0;
    s->up = ring ? 0 : (s->rank > 0 ? s->rank - 1 : -1);
    s->down = ring ? 0 : (s->rank < s->nranks - 1 ? s->rank + 1 : -1);
    s->ldims[0] = s->gt + s->own + s->gb;
    for (int k = 1; k < nd; ++k) s->ldims[k] = d->global_dims[k];
    if (int rc = make_plan(s->ldims, apps, &s->plan)) return rc;
    int got = 0;
    (void) lora_plan_get_option(s->plan, "steps_per_launch", &got);
    if (got != apps) {  // the local plan fuses differently from the probe (e.g. an odd innermost extent)
        s->apps = apps = got;
        s->fused = apps > 1;
        if (s->radius * apps > s->need) return LORA_EUNSUPPORTED;
    }
    const int brows = nd == 1 ? 4096 : (nd == 2 ? 32 : 1);
    s->strip = std::max(s->ghost, std::min(brows, s->own / 2));
    if (nd == 1) s->strip += s->strip & 1;
    s->esize = d->dtype == LORA_BF16 ? 2 : 8;
    s->bytes = lora_plan_padded_bytes(s->plan);
    s->row_bytes = s->bytes / (size_t) (s->ldims[0] + 2 * s->h0);
    SLAB_HIP(hipMalloc(&s->buf[0], s->bytes));
    SLAB_HIP(hipMalloc(&s->buf[1], s->bytes));
    SLAB_HIP(hipMemset(s->buf[0], 0, s->bytes));
    SLAB_HIP(hipMemset(s->buf[1], 0, s->bytes));
    SLAB_HIP(hipStreamCreateWithFlags(&s->cs, hipStreamNonBlocking));
    SLAB_HIP(hipStreamCreateWithFlags(&s->ms, hipStreamNonBlocking));
    SLAB_HIP(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    SLAB_HIP(hipEventCreateWithFlags(&s->ev_exch, hipEventDisableTiming));
    s->valid = s->ghost;
    *out = s.release();
    return LORA_OK;
}

int lora_slab_info(const lora_slab *s, lora_slab_info_t *info) {
    if (!s || !info) return LORA_EINVAL;
    std::memset(info, 0, sizeof *info);
    info->begin = s->begin;
    info->end = s->end;
    info->ghost = s->ghost;
    info->ghost_top = s->gt;
    info->ghost_bottom = s->gb;
    info->apps_per_launch = s->apps;
    info->exchange_every = s->every;
    info->steps_done = s->steps_done;
    for (int k = 0; k < s->nd; ++k) info->local_dims[k] = s->ldims[k];
    info->launches = s->launches;
    info->exchanges = s->exchanges;
    info->local_bytes = s->bytes;
    return LORA_OK;
}

void *lora_slab_buffer(lora_slab *s, int which) {
    if (!s) return nullptr;
    return which == 0 || which == 1 ? s->buf[which] : s->buf[s->cur];
}
void *lora_slab_stream(lora_slab *s) { return s ? s->cs : nullptr; }
lora_plan *lora_slab_plan(lora_slab *s) { return s ? s->plan : nullptr; }

static int reset_state(lora_slab *s) {
    s->steps_done = 0;
    s->cur = 0;
    s->valid = s->ghost;
    s->ringstate[0] = kRingInput;
    s->ringstate[1] = kRingZero;
    s->pending = false;
    return LORA_OK;
}

int lora_slab_load(lora_slab *s, const void *host_global_padded) {
    if (!s || !host_global_padded) return LORA_EINVAL;
    SLAB_HIP(hipSetDevice(s->device));
    SLAB_HIP(hipStreamSynchronize(s->cs));
    SLAB_HIP(hipStreamSynchronize(s->ms));
    const char *g = static_cast<const char *>(host_global_padded);
    const int rows = s->ldims[0] + 2 * s->h0;
    if (!s->ring) {
        // local padded row p = global padded row p + begin - gt: one contiguous piece
        SLAB_HIP(hipMemcpy(s->buf[0], g + (size_t) (s->begin - s->gt) * s->row_bytes, (size_t) rows * s->row_bytes,
                           hipMemcpyHostToDevice));
    } else {
        const int n0 = s->gdims[0];
        for (int p = 0; p < rows; ++p) {  // own rows + ghost rows taken modulo the global extent
            int gi = (s->begin - s->gt - s->h0 + p) % n0;
            if (gi < 0) gi += n0;
            SLAB_HIP(hipMemcpy(static_cast<char *>(s->buf[0]) + (size_t) p * s->row_bytes,
                               g + (size_t) (gi + s->h0) * s->row_bytes, s->row_bytes, hipMemcpyHostToDevice));
        }
    }
    SLAB_HIP(hipMemset(s->buf[1], 0, s->bytes));
    return reset_state(s);
}

int lora_slab_load_device(lora_slab *s, const void *d_local_padded) {
    if (!s || !d_local_padded) return LORA_EINVAL;
    SLAB_HIP(hipSetDevice(s->device));
    SLAB_HIP(hipStreamSynchronize(s->ms));
    SLAB_HIP(hipMemcpyAsync(s->buf[0], d_local_padded, s->bytes, hipMemcpyDeviceToDevice, s->cs));
    SLAB_HIP(hipMemsetAsync(s->buf[1], 0, s->bytes, s->cs));
    return reset_state(s);
}

int lora_slab_refresh_ghosts_many(lora_slab **ss, int n) {
    if (!ss || n < 1) return LORA_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (!ss[i]) return LORA_EINVAL;
        (void) hipSetDevice(ss[i]->device);
        if (int rc = flush(ss[i])) return rc;
    }
    if (int rc = exchange_all(ss, n, true)) return rc;
    for (int i = 0; i < n; ++i) {
        (void) hipSetDevice(ss[i]->device);
        if (int rc = flush(ss[i])) return rc;
        ss[i]->valid = ss[i]->ghost;
    }
    return LORA_OK;
}

int lora_slab_run_many(lora_slab **ss, int n, int times) {
    if (!ss || n < 1 || times < 0) return LORA_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (!ss[i]) return LORA_EINVAL;
        if (ss[i]->apps != ss[0]->apps || ss[i]->ghost != ss[0]->ghost || ss[i]->steps_done != ss[0]->steps_done)
            return LORA_EINVAL;  // slabs of one decomposition only
        if ((ss[i]->up >= 0 || ss[i]->down >= 0) && !ss[i]->have_comm) return LORA_EINVAL;
    }
    lora_slab *s0 = ss[0];
    int t = 0;
    while (t < times) {
        const bool even = s0->steps_done % 2 == 0;
        int napps = 1;
        if (s0->fused && even && times - t >= s0->apps)
            napps = s0->apps;
        else if (s0->fused && even && s0->nd >= 2 && s0->apps >= 4 && times - t >= 2)
            // the tail: four through the workgroup-row kernel where six leave four or five (the single-GPU driver's
            // choice: 944 against 2 x 885 us per launch at 16384^2), else two
            napps = (s0->nd == 2 && s0->apps == 6 && times - t >= 4) ? 4 : 2;
        if (int rc = launch_all(ss, n, napps)) return rc;
        t += napps;
    }
    return LORA_OK;
}

int lora_slab_run(lora_slab *s, int times) { return lora_slab_run_many(&s, 1, times); }
int lora_slab_refresh_ghosts(lora_slab *s) { return lora_slab_refresh_ghosts_many(&s, 1); }

int lora_slab_sync(lora_slab *s) {
    if (!s) return LORA_EINVAL;
    SLAB_HIP(hipSetDevice(s->device));
    if (int rc = flush(s)) return rc;
    SLAB_HIP(hipStreamSynchronize(s->cs));
    SLAB_HIP(hipStreamSynchronize(s->ms));
    return LORA_OK;
}

int lora_slab_store(lora_slab *s, void *host_global_padded) {
    if (!s || !host_global_padded) return LORA_EINVAL;
    if (int rc = lora_slab_sync(s)) return rc;
    // own rows; a slab at a global edge also returns its pad rows there (the driver's halo state); the left / right
    // (and y) pads travel with the rows
    const bool edge_top = s->up < 0 || s->ring, edge_bottom = s->down < 0 || s->ring;
    const int lo = s->h0 + s->gt - (edge_top ? s->h0 : 0);
    const int hi = s->h0 + s->gt + s->own + (edge_bottom ? s->h0 : 0);
    char *g = static_cast<char *>(host_global_padded);
    SLAB_HIP(hipMemcpy(g + (size_t) (lo + s->begin - s->gt) * s->row_bytes,
                       static_cast<const char *>(s->buf[s->cur]) + (size_t) lo * s->row_bytes,
                       (size_t) (hi - lo) * s->row_bytes, hipMemcpyDeviceToHost));
    return LORA_OK;
}

// The host-buffer operator on `ngpus` devices of this node (what the CLIs' --gpus N runs): one process, one slab per
// device, RCCL communicators from ncclCommInitAll.  LORA_SLAB_LOOPBACK=1 in the environment puts all slabs on device
// 0 with the in-process loopback exchange instead -- the rehearsal of this path on a one-GPU box (tests).
int lora_run_host_multi(int shape, int dtype, const void *in, void *out, const double *params, int times,
                        const int *dims, int ngpus, int quiet, lora_run_info *info) {
    if (!in || !out || !dims || times < 0 || ngpus < 1) return LORA_EINVAL;
    if (ngpus == 1) return lora_run_host_dtype(shape, dtype, in, out, params, times, dims, quiet, info);
    const int ndev = lora_device_count();
    if (ndev <= 0) {
        lora::set_last_error_text("no HIP device visible");
        return LORA_ENODEVICE;
    }
    const char *lb = std::getenv("LORA_SLAB_LOOPBACK");
    const bool loopback = lb && lb[0] == '1';
    if (!loopback && ngpus > ndev) {
        lora::set_last_error_text("more GPUs requested than this node has");
        return LORA_EINVAL;
    }
    const int nd = lora_shape_ndim(shape);
    if (nd == 0) return LORA_EINVAL;
    std::vector<lora_slab_comm> comms(ngpus);
    std::vector<void *> nccl(ngpus, nullptr);
    std::vector<lora_slab *> slabs(ngpus, nullptr);
    struct Cleanup {
        std::vector<lora_slab *> &s;
        std::vector<void *> &c;
        ~Cleanup() {
            for (lora_slab *x : s) lora_slab_destroy(x);
            for (void *x : c)
                if (x) (void) rccl().CommDestroy(x);
        }
    } cleanup{slabs, nccl};
    if (loopback) {
        if (int rc = lora_slab_comm_loopback(comms.data(), ngpus)) return rc;
    } else {
        if (!rccl().ok) {
            lora::set_last_error_text("librccl.so could not be loaded");
            return LORA_EUNSUPPORTED;
        }
        std::vector<int> devs(ngpus);
        for (int i = 0; i < ngpus; ++i) devs[i] = i;
        const int rc = rccl().CommInitAll(nccl.data(), ngpus, devs.data());
        if (rc) return rccl_fail("ncclCommInitAll", rc);
        for (int i = 0; i < ngpus; ++i)
            if (int r2 = lora_slab_comm_rccl(&comms[i], nccl[i])) return r2;
    }
    lora_slab_desc d{};
    d.shape = shape;
    d.dtype = dtype;
    for (int k = 0; k < nd; ++k) d.global_dims[k] = dims[k];
    d.params = params;
    d.nranks = ngpus;
    d.boundary = lora_set_default_boundary(LORA_BC_REFERENCE);
    (void) lora_set_default_boundary(d.boundary);
    if (d.boundary == LORA_BC_PERIODIC) {
        lora::set_last_error_text("--gpus N takes the reference and the Dirichlet boundary");
        return LORA_EUNSUPPORTED;
    }
    using clock = std::chrono::steady_clock;
    const auto t_total0 = clock::now();
    for (int r = 0; r < ngpus; ++r) {
        d.rank = r;
        d.device = loopback ? 0 : r;
        if (int rc = lora_slab_create(&slabs[r], &d, &comms[r])) return rc;
        if (int rc = lora_slab_load(slabs[r], in)) return rc;
    }
    // warm-up outside the timed region (the reference has none): one sweep, then the input again
    if (times > 0) {
        if (int rc = lora_slab_run_many(slabs.data(), ngpus, 1)) return rc;
        for (int r = 0; r < ngpus; ++r) {
            if (int rc = lora_slab_sync(slabs[r])) return rc;
            if (int rc = lora_slab_load(slabs[r], in)) return rc;
        }
    }
    const auto t0 = clock::now();
    if (int rc = lora_slab_run_many(slabs.data(), ngpus, times)) return rc;
    for (int r = 0; r < ngpus; ++r)
        if (int rc = lora_slab_sync(slabs[r])) return rc;
    const auto t1 = clock::now();
    // every cell of `out` is owned by exactly one slab (pads at the global edges by the edge slabs)
    for (int r = 0; r < ngpus; ++r)
        if (int rc = lora_slab_store(slabs[r], out)) return rc;
    const auto t_total1 = clock::now();
    double points = 1.0;
    for (int k = 0; k < nd; ++k) points *= dims[k];
    const size_t esize = dtype == LORA_BF16 ? 2 : 8;
    lora_run_info ri{};
    ri.sweep_seconds = std::chrono::duration<double>(t1 - t0).count();
    ri.total_seconds = std::chrono::duration<double>(t_total1 - t_total0).count();
    ri.gstencils = points * times / ri.sweep_seconds / 1e9;
    ri.gstencils_refconv = ri.gstencils * lora_shape_gstencil_factor(shape);
    ri.hbm_gbs = points * times * 2.0 * esize / ri.sweep_seconds / 1e9;
    ri.variant = LORA_VARIANT_DIRECT;
    lora_slab_info_t si;
    (void) lora_slab_info(slabs[0], &si);
    ri.steps_per_launch = si.apps_per_launch;
    lora::set_last_run_info(ri);
    if (info) *info = ri;
    if (!quiet) {
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
        std::printf("%s\n", lora::run_label(shape));
        std::printf("Time = %lld[ms]\n", (long long) std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
        std::printf("GStencil/s = %f\n", points * times * lora_shape_gstencil_factor(shape) / (us / 1e6) / 1e9);
        std::fflush(stdout);
    }
    return LORA_OK;
}

// The same operator on a Pa x Pb grid of blocks (group E), one block per device of this node.  LORA_SLAB_LOOPBACK=1: all
// blocks on device 0 with the loopback exchange (one-GPU rehearsal).  Reference boundary.
int lora_run_host_blocks(int shape, int dtype, const void *in, void *out, const double *params, int times, const int *dims,
                         const int *grid, int quiet, lora_run_info *info) {
    if (!in || !out || !dims || !grid || times < 0 || grid[0] < 1 || grid[1] < 1) return LORA_EINVAL;
    const int n = grid[0] * grid[1];
    const int nd = lora_shape_ndim(shape);
    if (nd != 2 && nd != 3) return LORA_EUNSUPPORTED;
    const int ndev = lora_device_count();
    if (ndev <= 0) {
        lora::set_last_error_text("no HIP device visible");
        return LORA_ENODEVICE;
    }
    const char *lb = std::getenv("LORA_SLAB_LOOPBACK");
    const bool loopback = lb && lb[0] == '1';
    if (!loopback && n > ndev) {
        lora::set_last_error_text("more GPUs requested than this node has");
        return LORA_EINVAL;
    }
    std::vector<lora_slab_comm> comms(n);
    std::vector<void *> nccl(n, nullptr);
    std::vector<lora_block *> blocks(n, nullptr);
    struct Cleanup {
        std::vector<lora_block *> &b;
        std::vector<void *> &c;
        ~Cleanup() {
            for (lora_block *x : b) lora_block_destroy(x);
            for (void *x : c)
                if (x) (void) rccl().CommDestroy(x);
        }
    } cleanup{blocks, nccl};
    if (n > 1) {
        if (loopback) {
            if (int rc = lora_slab_comm_loopback(comms.data(), n)) return rc;
        } else {
            if (!rccl().ok) {
                lora::set_last_error_text("librccl.so could not be loaded");
                return LORA_EUNSUPPORTED;
            }
            std::vector<int> devs(n);
            for (int i = 0; i < n; ++i) devs[i] = i;
            const int rc = rccl().CommInitAll(nccl.data(), n, devs.data());
            if (rc) return rccl_fail("ncclCommInitAll", rc);
            for (int i = 0; i < n; ++i)
                if (int r2 = lora_slab_comm_rccl(&comms[i], nccl[i])) return r2;
        }
    }
    lora_block_desc d{};
    d.shape = shape;
    d.dtype = dtype;
    for (int k = 0; k < nd; ++k) d.global_dims[k] = dims[k];
    d.grid[0] = grid[0];
    d.grid[1] = grid[1];
    d.params = params;
    using clock = std::chrono::steady_clock;
    const auto t_total0 = clock::now();
    for (int r = 0; r < n; ++r) {
        d.coords[0] = r / grid[1];
        d.coords[1] = r % grid[1];
        d.device = loopback ? 0 : r;
        if (int rc = lora_block_create(&blocks[r], &d, n > 1 ? &comms[r] : nullptr)) return rc;
        if (int rc = lora_block_load(blocks[r], in)) return rc;
    }
    if (times > 0) {  // warm-up outside the timed region, then the input again
        if (int rc = lora_block_run_many(blocks.data(), n, 1)) return rc;
        for (int r = 0; r < n; ++r) {
            if (int rc = lora_block_sync(blocks[r])) return rc;
            if (int rc = lora_block_load(blocks[r], in)) return rc;
        }
    }
    const auto t0 = clock::now();
    if (int rc = lora_block_run_many(blocks.data(), n, times)) return rc;
    for (int r = 0; r < n; ++r)
        if (int rc = lora_block_sync(blocks[r])) return rc;
    const auto t1 = clock::now();
    for (int r = 0; r < n; ++r)
        if (int rc = lora_block_store(blocks[r], out)) return rc;
    const auto t_total1 = clock::now();
    double points = 1.0;
    for (int k = 0; k < nd; ++k) points *= dims[k];
    const size_t esize = dtype == LORA_BF16 ? 2 : 8;
    lora_run_info ri{};
    ri.sweep_seconds = std::chrono::duration<double>(t1 - t0).count();
    ri.total_seconds = std::chrono::duration<double>(t_total1 - t_total0).count();
    ri.gstencils = points * times / ri.sweep_seconds / 1e9;
    ri.gstencils_refconv = ri.gstencils * lora_shape_gstencil_factor(shape);
    ri.hbm_gbs = points * times * 2.0 * esize / ri.sweep_seconds / 1e9;
    ri.variant = LORA_VARIANT_DIRECT;
    lora_block_info_t bi;
    (void) lora_block_info(blocks[0], &bi);
    ri.steps_per_launch = bi.apps_per_launch;
    lora::set_last_run_info(ri);
    if (info) *info = ri;
    if (!quiet) {
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
        std::printf("%s\n", lora::run_label(shape));
        std::printf("Time = %lld[ms]\n", (long long) std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
        std::printf("GStencil/s = %f\n", points * times * lora_shape_gstencil_factor(shape) / (us / 1e6) / 1e9);
        std::fflush(stdout);
    }
    return LORA_OK;
}

}  // extern "C"
