"""Multi-GPU slab decomposition of the sweep: one process per GPU, halo exchange over RCCL (xGMI).

The reference is single-GPU (no NCCL/MPI call sites); this is the multi-GPU form the north star
asks for.  The outermost interior dimension (rows in 2D, planes in 3D, points in 1D) is cut into contiguous
slabs, one per rank of a ``torch.distributed`` process group (backend "nccl" = RCCL on ROCm).

Layout of a rank's local array (same padded layout as the single-GPU operator, so the slab kernels ARE the
single-GPU kernels)::

      [ pad ][ G ghost rows of the upper neighbour ][ ml own rows ][ G ghost rows of the lower one ][ pad ]

* Ghost rows are ordinary interior rows of the local problem.  A launch that advances the grid by ``need`` rows of
  reach (radius x applications per launch: 3 or 6 in 2D, 1 or 2 in 3D, 4 or 32 in 1D) is run on the own rows plus the ghost rows
  that are still needed later, after which ``need`` fewer ghost rows are valid.  Ghost zones are G = need x E deep and
  are refreshed from the neighbours' own rows every E launches (communication-avoiding: E x fewer, E x larger
  messages -- at 8 GPUs a 2048 x 16384 slab sweeps in ~65 us, so per-launch host work and P2P latency, not
  bandwidth, are what has to be amortised).  Redundant work: E(E-1)/2 x need ghost rows per side per period.
* On the launch that exhausts the ghost zone the two own boundary strips are swept first, their exchange is posted
  (it runs on the process group's own stream) and the interior is swept while the messages are in flight.
* A rank at a global edge has no ghost rows there: its pad rows are the global halo and keep the reference's
  semantics untouched (never written: the caller's input halo at even time levels, zeros at odd ones);
  ``boundary="dirichlet"`` keeps the caller's halo at every level instead; ``boundary="periodic"`` closes the slabs
  into a ring (the first and the last rank exchange ghost rows too) and wraps the unsplit dimensions locally before
  every sweep.
* Fused launches (``Plan.stepk_region``: 2 applications in 2D / 3D, 8 in 1D) are used exactly like in the
  single-GPU driver: both physical buffers then carry the level-0 halo ring, tails are single sweeps.

Per-point arithmetic is identical to the single-GPU sweep (same kernels, same tap order), so an N-rank result is
bit-identical to the 1-rank result.

The sweep itself is delegated to a *stepper* (``step_region`` / ``stepk_region``).  The product stepper is
``HipStepper`` (the HIP engine through ``ops.Plan``); there is no CPU stepper in this package -- the gloo tests
inject one to exercise the decomposition, ghost-zone bookkeeping and exchange logic on CPU.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass
from typing import Callable, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import ops


@dataclass(frozen=True)
class SlabLayout:
    """Which part of the global grid a rank owns (outermost interior dimension only) and its ghost depth."""

    shape: int
    global_dims: tuple
    world_size: int
    rank: int
    begin: int  # first global interior index of the slab
    end: int  # one past the last
    ghost: int = 0  # ghost rows per neighbour side
    ring: bool = False  # periodic along the split dimension: the first and the last slab are neighbours too

    @property
    def own(self) -> int:
        return self.end - self.begin

    @property
    def ghost_top(self) -> int:
        return self.ghost if (self.rank > 0 or self.ring) else 0

    @property
    def ghost_bottom(self) -> int:
        return self.ghost if (self.rank < self.world_size - 1 or self.ring) else 0

    @property
    def local_dims(self) -> tuple:
        return (self.ghost_top + self.own + self.ghost_bottom,) + tuple(self.global_dims[1:])

    @property
    def halo0(self) -> int:
        return ops.halo(self.shape)[0]

    @property
    def halos(self) -> tuple:
        """Pad cells per side of every dimension of the padded layout."""
        return tuple(ops.halo(self.shape))

    @property
    def radius0(self) -> int:
        """Stencil reach along the split dimension per application."""
        return {1: 4, 2: 3, 3: 1}[len(self.global_dims)]


def slab_layout(shape, global_dims: Sequence[int], world_size: int, rank: int, multiple: int | None = None,
                ghost: int = 0, ring: bool = False) -> SlabLayout:
    """Balanced contiguous split; every slab boundary is a multiple of ``multiple`` (default: 32 rows in 2D,
    2 points in 1D, 1 plane in 3D)."""
    sid = ops.shape_id(shape)
    nd = len(global_dims)
    if multiple is None:
        multiple = {1: 2, 2: 32, 3: 1}[nd]
    n0 = int(global_dims[0])
    units = -(-n0 // multiple)  # ceil
    if units < world_size:
        raise ValueError(f"cannot split {n0} into {world_size} slabs of multiples of {multiple}")
    base, extra = divmod(units, world_size)
    start_u = rank * base + min(rank, extra)
    end_u = start_u + base + (1 if rank < extra else 0)
    begin, end = start_u * multiple, min(end_u * multiple, n0)
    lay = SlabLayout(sid, tuple(int(d) for d in global_dims), world_size, rank, begin, end,
                     ghost if (world_size > 1 or ring) else 0, ring)
    if (world_size > 1 or ring) and end - begin < max(lay.radius0, ghost):
        raise ValueError("slab thinner than its ghost zone")
    return lay


def next_depth(ndim: int, apps: int, remaining: int, even: bool, fused: bool = True, tails=(2, 4)) -> int:
    """Applications of the next launch of a slab / block run whose full launches fuse `apps`: the full depth at an even
    time level while that many sweeps remain; then (2D / 3D, apps >= 4) a tail launch -- four through the workgroup-row
    kernel where a six-application 2D run leaves four or five (944 against 2 x 885 us per launch at 16384^2), else two;
    single sweeps otherwise.  `tails`: the tail depths the stepper has.  ONE rule for slab.py, blocks.py, bench.py; slab.cpp
    and blocks.cpp (launch_all) state the same rule in C++."""
    if fused and even and apps > 1 and remaining >= apps:
        return apps
    if fused and even and ndim >= 2 and apps >= 4 and remaining >= 2:
        if ndim == 2 and apps == 6 and remaining >= 4 and 4 in tails:
            return 4
        if 2 in tails:
            return 2
    return 1


def launch_depths(ndim: int, apps: int, times: int, fused: bool = True, tails=(2, 4), steps_done: int = 0) -> list:
    """The launch depths of a whole run of `times` sweeps, in order (next_depth applied until none remain)."""
    out, t = [], 0
    while t < times:
        d = next_depth(ndim, apps, times - t, (steps_done + t) % 2 == 0, fused, tails)
        out.append(d)
        t += d
    return out


def stepper_tails(stepper) -> tuple:
    """Tail depths a stepper can launch under a deeper plan: every depth for one with stepn_region (the HIP plans), two for
    one with step2_region only (test steppers)."""
    if hasattr(stepper, "stepn_region"):
        return (2, 4)
    return (2,) if hasattr(stepper, "step2_region") else ()


def slab_schedule(sid, global_dims, world_size, rank, ring, make_stepper, fused, exchange_every, all_agree=bool):
    """Launch depth, ghost need and refresh interval of a slab decomposition, and this rank's layout and stepper.
    Everything that fixes the exchange pattern comes from rank-independent data; `all_agree` ANDs a local verdict over
    the ranks (identity for a single process).  Returns (layout, stepper, apps, need, every) or None when even single
    sweeps do not fit the thinnest slab."""
    nd = len(global_dims)
    radius = slab_layout(sid, global_dims, world_size, rank, ring=ring).radius0
    # thinnest slab of the decomposition bounds the ghost depth (neighbours supply ghost rows from own rows)
    thinnest = min(slab_layout(sid, global_dims, world_size, r).own for r in range(world_size))
    auto_every = exchange_every is None
    split = world_size > 1 or ring
    # Applications of a fused launch (lora_plan_stepk): fixed ONCE from rank-independent data -- a probe stepper on the
    # GLOBAL dims, as csrc/slab.cpp does -- and then forced on every rank's local plan.  (A plan resolves its depth from
    # its own grid size: a 3D fp64 plan takes four per launch from ~1e7 points and two below, and the ranks at the ends
    # of the decomposition hold fewer planes than the ones in the middle.  Ranks that resolved different depths would
    # disagree on ghost depth, message sizes and exchange cadence: star3d1r 208 x 384 x 384 on 4 ranks gave (2, 4) at
    # the ends and (4, 8) in the middle.)
    default_apps = 8 if nd == 1 else 2
    probe_apps = 1
    if fused:
        whole = make_stepper(slab_layout(sid, global_dims, 1, 0))
        if getattr(whole, "wants_fused", True) and (hasattr(whole, "stepk_region") or hasattr(whole, "step2_region")):
            probe_apps = int(getattr(whole, "apps_per_launch", default_apps))
        del whole
    candidates = [probe_apps]
    while candidates[-1] > 1:  # what a slab too thin for the probe's depth falls back to: csrc/slab.cpp's sequence
        a = candidates[-1]
        candidates.append(a - 2 if (nd >= 2 and a >= 4) else 1)
    for apps in candidates:
        need = radius * apps
        if split and thinnest < need:
            continue
        if auto_every:
            # refresh as rarely as keeps the redundant ghost sweeps (about (E - 1) x need rows per launch) within
            # ~10 % of a slab: 8 launches for the 2D / 1D configurations, 4 for the thin 3D slabs of an 8-GPU run
            # (ring of one over RCCL, 2048 x 16384 slab: E = 2 / 4 / 8 / 16 -> 492 / 535 / 553 / 534 GStencils/s)
            exchange_every = 8
            while exchange_every > 1 and (exchange_every - 1) * need > 0.1 * thinnest:
                exchange_every //= 2
        e = max(1, min(exchange_every, thinnest // need if split else exchange_every))
        layout = slab_layout(sid, global_dims, world_size, rank, ghost=need * e, ring=ring)
        stepper = make_stepper(layout)
        agrees = True
        if apps > 1:
            if hasattr(stepper, "set_apps"):
                stepper.set_apps(apps)  # the local plan takes the depth it is given, whatever its own size says
            has = hasattr(stepper, "stepk_region") or hasattr(stepper, "step2_region")
            agrees = bool(has and getattr(stepper, "wants_fused", True)
                          and int(getattr(stepper, "apps_per_launch", default_apps)) == apps)
        # every rank takes the same branch: one rank's local plan declining the depth (an odd innermost extent, a
        # stepper without fused launches) moves ALL ranks on to the next candidate, never just that one
        if not all_agree(agrees):
            continue
        return layout, stepper, apps, need, e
    return None


class _GatherWork:
    """Completion handle of the all-gather form of the ghost exchange: wait() also scatters the neighbours' strips."""

    def __init__(self, work, everyone, t, g, first, last, up, down):
        self.work, self.everyone, self.t, self.g = work, everyone, t, g
        self.first, self.last, self.up, self.down = first, last, up, down

    def wait(self):
        self.work.wait()
        g = self.g
        if self.up is not None:
            self.t[self.first - g:self.first].copy_(self.everyone[self.up][g:2 * g])  # its bottom strip
        if self.down is not None:
            self.t[self.last:self.last + g].copy_(self.everyone[self.down][:g])       # its top strip
        return True


class HipStepper:
    """Product stepper: the HIP engine on the local array (own rows + ghost rows)."""

    def __init__(self, layout: SlabLayout, params=None, weights=None, dtype="f64", boundary="reference",
                 options=None, variant=None):
        self.plan = ops.Plan(layout.shape, layout.local_dims, params, dtype=dtype)
        if weights is not None:
            self.plan.set_weights(weights)
        # kernel options / variant come BEFORE the driver fixes its ghost depth: they decide how many applications a
        # launch fuses (apps_per_launch), which the layout is built around
        if variant is not None:
            self.plan.set_variant(variant)
        for k, v in (options or {}).items():
            self.plan.set_option(k, int(v))
        if boundary == "dirichlet":
            self.plan.set_boundary(boundary)  # fused launches: intermediate halo cells keep the source's values
        # (3D slabs keep the plan's four sweeps per launch whatever their thickness: with the second form of the
        # register-resident kernel, its chunk-length model and whole-slab launches a 64-plane share of star3d1r 512^3 runs
        # 479-521 GStencils/s per rank with four per launch against 357-380 with two -- tools/cslab_3d_k.py; the first form
        # lost to the two-sweep kernels below 96 planes)
        if boundary == "periodic":
            self.plan.set_option("steps_per_launch", 1)  # a fused launch would need the wrap of its inner levels
        elif len(layout.local_dims) == 3 and self.plan.get_option("steps_per_launch") == 3:
            # the single-GPU plane-streaming schedule fuses three sweeps per launch on the reference's alternating buffer
            # state; slab launches start at even steps on buffers that both carry the halo: two per launch (or the four of
            # the register-resident kernel, which the plan resolves by itself on big fp64 grids)
            self.plan.set_option("steps_per_launch", 2)
        # launches go to the stream that is current when the driver is built (looked up once: at 8 GPUs a launch is
        # ~100 us of GPU time, so per-call host work matters)
        self.torch_stream = torch.cuda.current_stream() if torch.cuda.is_available() else None
        self.stream = int(self.torch_stream.cuda_stream) if self.torch_stream is not None else 0

    @property
    def apps_per_launch(self) -> int:
        """Applications of one fused launch: 2 in 2D / 3D, 8 in 1D (the plan's resolved steps_per_launch)."""
        return self.plan.get_option("steps_per_launch")

    @property
    def wants_fused(self) -> bool:
        return self.apps_per_launch > 1

    def set_apps(self, apps: int) -> None:
        """Force the launch depth the slab driver chose from the GLOBAL grid (every rank the same), whatever this rank's
        local grid size would have resolved; apps_per_launch then says what the plan really does."""
        self.plan.set_option("steps_per_launch", int(apps))

    def step_region(self, src: torch.Tensor, dst: torch.Tensor, begin: int, end: int) -> None:
        self.plan.step_region(src.data_ptr(), dst.data_ptr(), begin, end, stream=self.stream)

    def stepk_region(self, src: torch.Tensor, dst: torch.Tensor, begin: int, end: int) -> None:
        self.plan.stepk_region(src.data_ptr(), dst.data_ptr(), begin, end, stream=self.stream)

    def step2_region(self, src: torch.Tensor, dst: torch.Tensor, begin: int, end: int) -> None:
        self.plan.step2_region(src.data_ptr(), dst.data_ptr(), begin, end, stream=self.stream)

    def stepn_region2(self, napps: int, src: torch.Tensor, dst: torch.Tensor, b0: int, e0: int, b1: int, e1: int) -> None:
        """Two disjoint ranges in one call (lora_plan_stepn_region2): the two ends behind a deferred wait."""
        self.plan.stepn_region2(napps, src.data_ptr(), dst.data_ptr(), b0, e0, b1, e1, stream=self.stream)

    def stepn_region(self, napps: int, src: torch.Tensor, dst: torch.Tensor, begin: int, end: int) -> None:
        """A tail launch of `napps` applications under a deeper plan (lora_plan_stepn_region)."""
        self.plan.stepn_region(napps, src.data_ptr(), dst.data_ptr(), begin, end, stream=self.stream)

    def wrap(self, buf: torch.Tensor) -> None:
        """Periodic halo of the LOCAL array (lora_plan_halo, wrap mode): right along the unsplit dimensions; along
        the split one it only touches pad rows beyond the ghost zones, which nothing reads."""
        self.plan.halo(buf.data_ptr(), "wrap", stream=self.stream)


class SlabDriver:
    """Time-step driver of one rank's slab (reference driver semantics, 2d/gpu.cu:525-554, per slab)."""

    def __init__(self, shape, global_dims: Sequence[int], group=None, device=None, params=None, weights=None,
                 stepper_factory: Callable[[SlabLayout], object] | None = None, overlap: bool | None = None,
                 exchange_every: int | None = None, fused: bool | None = None, boundary_rows: int | None = None,
                 dtype="f64", boundary: str = "reference", ring_of_one: bool = False, options=None, variant=None):
        if boundary not in ("reference", "dirichlet", "periodic"):
            raise ValueError("boundary must be reference, dirichlet or periodic")
        self.dirichlet = boundary == "dirichlet"
        self.periodic = boundary == "periodic"
        if self.periodic:
            fused = False  # single sweeps, each preceded by the wrap of the unsplit dimensions
        # ring_of_one: a single rank is treated as a ring of one slab that exchanges its ghost rows with ITSELF through
        # the process group -- the complete exchange path (P2P batch, overlap, waits) over the real RCCL backend on a
        # one-GPU box.  With the periodic boundary the result is the torus (tested against the oracle); with the other
        # boundaries it is a TIMING rehearsal of one rank's share of an N-GPU run (tools/slab_overhead.py), not a result.
        self.group = group
        self._host_side_p2p = dist.is_initialized() and dist.get_backend(group) != "nccl"
        self.exchange_mode = os.environ.get("LORA_SLAB_EXCHANGE", "p2p")  # "p2p" (default) or "allgather"
        if self.exchange_mode not in ("p2p", "allgather"):
            raise ValueError("LORA_SLAB_EXCHANGE must be p2p or allgather")
        self._p2p_worked = False
        self._pending = []       # works of a ghost exchange the next launch has not waited for yet
        self.defer_wait = os.environ.get("LORA_SLAB_DEFER_WAIT", "1") != "0"
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._ring = (self.periodic and self.world_size > 1) or (ring_of_one and self.world_size == 1
                                                                 and dist.is_initialized())
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        sid = ops.shape_id(shape)
        nd = len(global_dims)
        self.ndim = nd
        # a first, ghost-free layout tells how thick the slabs are; the ghost depth is then fitted to them
        probe = slab_layout(sid, global_dims, self.world_size, self.rank, ring=self._ring)
        radius = probe.radius0
        if fused is None:
            fused = True  # refined below by what the stepper supports
        self.dtype = ops.dtype_id(dtype)
        self.torch_dtype = torch.bfloat16 if self.dtype == ops.DTYPES["bf16"] else torch.float64
        self._make_stepper = stepper_factory or (
            lambda lay: HipStepper(lay, params=params, weights=weights, dtype=self.dtype, boundary=boundary,
                                   options=options, variant=variant))
        sched = slab_schedule(sid, global_dims, self.world_size, self.rank, self._ring, self._make_stepper, fused,
                              exchange_every, self._all_agree)
        layout = None
        if sched is not None:
            layout, self.stepper, self.apps, self.need, self.exchange_every = sched
            self.fused = self.apps > 1
        if layout is None or not hasattr(self, "stepper"):
            raise ValueError("slabs are thinner than the stencil radius")
        self.layout = layout
        self.radius = radius
        # boundary strips first: the default in 1D only; in 2D and 3D the whole slab goes in one launch (csrc/slab.cpp: the
        # fused kernels run ONE round of workgroups sized to their region, a thin strip costs a third of a whole-slab
        # launch) and only the deferred wait hides the link
        self.overlap = (nd == 1) if overlap is None else bool(overlap)
        self.local_padded_shape = ops.padded_shape(sid, layout.local_dims)
        self.buf = [torch.zeros(self.local_padded_shape, dtype=self.torch_dtype, device=self.device) for _ in range(2)]
        if boundary_rows is None:
            boundary_rows = {1: 4096, 2: 32, 3: 1}[nd]
        self.strip = max(layout.ghost, min(boundary_rows, layout.own // 2))
        if nd == 1:
            self.strip += self.strip & 1  # 1D regions start on even points
        self.up = self.rank - 1 if self.rank > 0 else None  # neighbour owning smaller indices
        self.down = self.rank + 1 if self.rank < self.world_size - 1 else None
        if self._ring:  # the first and the last slab are neighbours (a ring of one: its own neighbour)
            self.up = (self.rank - 1) % self.world_size
            self.down = (self.rank + 1) % self.world_size
        self.steps_done = 0
        self.cur = 0  # physical buffer holding the current time level
        self.valid = layout.ghost  # ghost rows per side that hold the current time level
        self.ring = ["input", "zero"]  # what the halo ring of each physical buffer holds (fused-launch bookkeeping)

    def _all_agree(self, ok: bool) -> bool:
        """AND of a local verdict over the ranks of the group (a collective: every rank calls it at the same point of
        the constructor), so that no rank raises or moves on alone while the others enter an exchange and hang."""
        if not (dist.is_initialized() and self.world_size > 1):
            return bool(ok)
        dev = self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def _on_stream(self):
        """Everything the driver enqueues -- sweeps, halo-ring copies, the P2P batch and its waits -- must be ordered on
        ONE stream: the stepper launches on the stream that was current when it was built, while torch copies and
        `work.wait()` (NCCL: a stream-side wait) act on the stream that is current NOW.  Make them the same."""
        ts = getattr(self.stepper, "torch_stream", None)
        return torch.cuda.stream(ts) if ts is not None else contextlib.nullcontext()

    # ---- data movement between the global padded array and the slabs ---------------------------------
    def load_global(self, global_padded) -> None:
        """buffer 0 <- this rank's rows of the global padded input (ghost rows and pads included), buffer 1 <- 0."""
        lay = self.layout
        h0 = lay.halo0
        if lay.ring:
            # own rows + ghost rows taken modulo the global extent; the pad rows beyond them are never read
            n0 = lay.global_dims[0]
            rows = (np.arange(lay.begin - lay.ghost_top - h0, lay.end + lay.ghost_bottom + h0) % n0) + h0
            part = global_padded[torch.from_numpy(rows)] if isinstance(global_padded, torch.Tensor) \
                else np.ascontiguousarray(np.asarray(global_padded)[rows])
        else:
            lo = lay.begin - lay.ghost_top
            hi = lay.end + lay.ghost_bottom + 2 * h0
            part = global_padded[lo:hi]
        if isinstance(part, np.ndarray):
            part = torch.from_numpy(np.ascontiguousarray(part))
        with self._on_stream():
            self._flush()
            self.buf[0].copy_(part.to(self.device))
            self.buf[1].zero_()
        self.steps_done, self.cur, self.valid = 0, 0, lay.ghost
        self.ring = ["input", "zero"]

    def load_local(self, local_padded: torch.Tensor) -> None:
        """Same from an already-local array (own rows + ghost rows + pads), e.g. generated on the device."""
        with self._on_stream():
            self._flush()
            self.buf[0].copy_(local_padded)
            self.buf[1].zero_()
        self.steps_done, self.cur, self.valid = 0, 0, self.layout.ghost
        self.ring = ["input", "zero"]

    def result(self) -> torch.Tensor:
        """The local padded buffer holding the current time level (its producers are ordered on the stepper's stream:
        a consumer on another stream synchronises with it first)."""
        with self._on_stream():
            self._flush()
        return self.buf[self.cur]

    def gather_global(self, dst_rank: int = 0):
        """Assemble the global padded result on ``dst_rank`` (own rows of every slab, global-edge halos from the
        edge ranks; left/right halos travel with the rows).  Returns a CPU tensor there, None elsewhere."""
        lay = self.layout
        h0 = lay.halo0
        with self._on_stream():
            self._flush()
            if self.periodic:
                self.stepper.wrap(self.buf[self.cur])  # the result is a consistent periodic array (like lora_plan_run)
        cur = self.result()
        ts = getattr(self.stepper, "torch_stream", None)
        if ts is not None:
            ts.synchronize()  # the copy to the host below runs on whatever stream is current
        edge_top = self.up is None or (self.periodic and self.rank == 0)
        edge_bottom = self.down is None or (self.periodic and self.rank == self.world_size - 1)
        lo = h0 + lay.ghost_top - (h0 if edge_top else 0)
        hi = h0 + lay.ghost_top + lay.own + (h0 if edge_bottom else 0)
        piece = cur[lo:hi].cpu()
        if self.world_size == 1:
            if self._ring:  # a ring of one: the pad rows of the split dimension = its own opposite interior edge
                n0 = lay.global_dims[0]
                piece[:h0] = piece[n0:n0 + h0].clone()
                piece[h0 + n0:] = piece[h0:2 * h0].clone()
            return piece
        pieces = [None] * self.world_size if self.rank == dst_rank else None
        dist.gather_object(piece, pieces, dst=dst_rank, group=self.group)
        if self.rank != dst_rank:
            return None
        full = torch.cat(pieces, dim=0)
        if self.periodic:  # pad rows of the split dimension = the opposite interior edge
            n0 = lay.global_dims[0]
            full[:h0] = full[n0:n0 + h0].clone()
            full[h0 + n0:] = full[h0:2 * h0].clone()
        return full

    # ---- halo-ring bookkeeping of the fused path -----------------------------------------------------
    def _set_ring(self, b: int, what: str, src: int) -> None:
        """Global-edge halo ring of physical buffer b: 'input' (copy of buffer src's ring) or 'zero'."""
        if self.ring[b] == what:
            return
        t = self.buf[b]
        s = self.buf[src]
        for d, k in enumerate(self.layout.halos):  # 2D: 4, 4; 3D: 1, 2, 4 cells on either side
            for side in (slice(0, k), slice(-k, None)):
                idx = (slice(None),) * d + (side,)
                if what == "zero":
                    t[idx].zero_()
                else:
                    t[idx].copy_(s[idx])
        self.ring[b] = what

    # ---- ghost exchange ---------------------------------------------------------------------------------
    def _post_exchange(self, t: torch.Tensor):
        """Own boundary rows of ``t`` -> the neighbours' ghost zones (G rows each way)."""
        lay = self.layout
        g = lay.ghost
        if t.is_cuda and self._host_side_p2p:
            # gloo moves device tensors from a CPU thread with no stream ordering (rehearsals of the N > 1 path on one
            # GPU): the strips just launched must have landed before it reads them.  RCCL is stream-ordered.
            torch.cuda.current_stream(t.device).synchronize()
        first = lay.halo0 + lay.ghost_top  # padded index of the first own row
        last = first + lay.own
        if self.exchange_mode == "p2p":
            # order: in a ring of two both neighbours are the same peer, and backends match the k-th send to a peer
            # with its k-th receive from us -- our top strip must meet ITS bottom ghost zone
            opsl = []
            if self.up is not None:
                opsl.append(dist.P2POp(dist.isend, t[first:first + g], self.up, group=self.group))
            if self.down is not None:
                opsl.append(dist.P2POp(dist.irecv, t[last:last + g], self.down, group=self.group))
                opsl.append(dist.P2POp(dist.isend, t[last - g:last], self.down, group=self.group))
            if self.up is not None:
                opsl.append(dist.P2POp(dist.irecv, t[first - g:first], self.up, group=self.group))
            try:
                return dist.batch_isend_irecv(opsl) if opsl else []
            except RuntimeError:
                if self._p2p_worked:
                    raise
                self.exchange_mode = "allgather"  # point-to-point refused by this backend / topology: collectives only
        # Fallback (LORA_SLAB_EXCHANGE=allgather, or P2P refused on first use): every rank contributes its two boundary
        # strips to one all-gather and picks its neighbours' -- N x the bytes of the P2P form, collectives only.
        mine = torch.cat([t[first:first + g], t[last - g:last]], dim=0).contiguous()
        flat = torch.empty((self.world_size * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype,
                           device=mine.device)  # concatenated along dim 0 (the layout every backend accepts)
        work = dist.all_gather_into_tensor(flat, mine, group=self.group, async_op=True)
        everyone = flat.view((self.world_size,) + tuple(mine.shape))
        return [_GatherWork(work, everyone, t, g, first, last, self.up, self.down)]

    def refresh_ghosts(self) -> None:
        """Blocking refresh of the ghost zones of the current buffer (used after loading device-generated data)."""
        with self._on_stream():
            self._flush()
            if self.up is not None or self.down is not None:
                for w in self._post_exchange(self.buf[self.cur]):
                    w.wait()
                self._p2p_worked = True
        self.valid = self.layout.ghost

    # ---- one launch (1 or 2 applications) ---------------------------------------------------------------
    def _launch(self, napps: int) -> None:
        """One launch of `napps` applications: the driver's own fused count (stepk_region), a tail depth of a run whose
        launches fuse four or six (stepn_region / step2_region: next_depth) or 1 (step_region)."""
        lay = self.layout
        fused = napps > 1
        apps = napps
        need = self.radius * apps
        src_i, dst_i = self.cur, 1 - self.cur
        src, dst = self.buf[src_i], self.buf[dst_i]
        if fused and napps == self.apps:
            sweep = getattr(self.stepper, "stepk_region", None) or self.stepper.step2_region
        elif fused and hasattr(self.stepper, "stepn_region"):
            stepn = self.stepper.stepn_region

            def sweep(a, b, lo, hi):
                stepn(napps, a, b, lo, hi)
        elif fused:
            if napps != 2:
                raise ValueError(f"this stepper has no launch of {napps} applications")
            sweep = self.stepper.step2_region
        else:
            sweep = self.stepper.step_region
        if self.periodic:
            self._flush()  # the wrap writes the x / y halo of every row, ghost rows in flight included
            self.stepper.wrap(src)  # x / y (and, on one rank, the split dimension too): halo = opposite interior edge
        else:
            ring = None
            if self.dirichlet:
                ring = "input"  # fixed boundary: every level carries the caller's halo ring
            elif self.fused or fused:
                # fused launches need the level-0 ring in both buffers; a single sweep from an even level writes the
                # odd level, whose ring is 0; from an odd level it writes an even one (ring = input)
                even = self.steps_done % 2 == 0
                ring = "input" if (fused or not even) else "zero"
            if ring is not None and self.ring[dst_i] != ring:
                self._flush()  # the ring copy reads the source's halo columns, ghost rows included
                self._set_ring(dst_i, ring, src_i)
        gt, own = lay.ghost_top, lay.own
        if self.up is None and self.down is None:
            sweep(src, dst, 0, own)
        else:
            assert self.valid >= need, "ghost zone exhausted"
            left = self.valid - need  # ghost rows still valid after this launch
            exchange = left < self.need  # not enough for another launch of this driver's kind: refresh now
            if exchange:
                self._flush()  # (only when every launch exchanges)
                s = self.strip
                if self.overlap and own > 2 * s:
                    if self.up is not None:
                        sweep(src, dst, gt, gt + s)
                    if self.down is not None:
                        sweep(src, dst, gt + own - s, gt + own)
                    works = self._post_exchange(dst)
                    sweep(src, dst, gt + (s if self.up is not None else 0), gt + own - (s if self.down is not None else 0))
                else:
                    sweep(src, dst, gt, gt + own)
                    works = self._post_exchange(dst)
                # The ghost rows in flight are first read by the NEXT launch, and only by the part of it within
                # `need` rows of the slab's ends: leave the messages pending -- the next launch sweeps its deep interior
                # before it waits for them, which doubles the compute the transfer can hide behind.
                self._pending = list(works)
                if not self.defer_wait:
                    self._flush()
                self.valid = lay.ghost
            else:
                lo = gt - (left if self.up is not None else 0)
                hi = gt + own + (left if self.down is not None else 0)
                a, b = gt + need, gt + own - need  # output rows whose inputs are own rows only
                if self._pending and b - a >= 2 * need:
                    sweep(src, dst, a, b)
                    self._flush()
                    if hasattr(self.stepper, "stepn_region2"):  # both ends in one call (one launch for the 3D lanes kernels)
                        self.stepper.stepn_region2(napps, src, dst, lo, a, b, hi)
                    else:
                        sweep(src, dst, lo, a)
                        sweep(src, dst, b, hi)
                else:
                    self._flush()
                    sweep(src, dst, lo, hi)
                self.valid = left
        self.cur = dst_i
        self.steps_done += apps

    def _flush(self) -> None:
        """Wait for a ghost exchange left in flight by the previous launch."""
        if self._pending:
            for w in self._pending:
                w.wait()
            self._pending = []
            self._p2p_worked = True

    def step(self) -> None:
        """One kernel application."""
        with self._on_stream():
            self._launch(1)

    def run(self, times: int) -> None:
        """`times` kernel applications (fused pairs where the shape allows, starting at even time levels)."""
        t = 0
        tails = stepper_tails(self.stepper)
        with self._on_stream():
            while t < times:
                d = next_depth(self.ndim, self.apps, times - t, self.steps_done % 2 == 0, self.fused, tails)
                self._launch(d)
                t += d
