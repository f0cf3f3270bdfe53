"""Multi-rank slab path on CPU: world_size 2 and 3 over gloo.  The decomposition, the halo exchange (including
the boundary-first ordering and the untouched global-edge halos) and the gather are the product code of
lorastencil_amd/slab.py; only the per-slab sweep is replaced by an oracle-backed stepper, because the HIP
stepper needs a GPU.  The N-rank result must equal the 1-rank oracle result bit-for-bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleStepper:
    """CPU stand-in for HipStepper (tests only)."""

    def __init__(self, layout, weights, wants_fused=True, dirichlet=False, apps=None):
        from lorastencil_amd import ops

        self.dirichlet = dirichlet
        self.shape = layout.shape
        self.w = weights
        self.h = ops.halo(layout.shape)
        self.wants_fused = wants_fused
        self.apps_per_launch = apps or (8 if len(self.h) == 1 else 2)
        self.calls = {"step": 0, "step2": 0, "two": 0}

    def step2_region(self, src, dst, begin, end):
        """Two applications (lora_plan_step2_region): the tail launches of a driver that fuses four."""
        self.calls["two"] += 1
        self.stepk_region(src, dst, begin, end, apps=2)

    def stepk_region(self, src, dst, begin, end, apps=None):
        """apps_per_launch applications in one call (lora_plan_stepk_region): cells outside the local interior are 0 at
        odd intermediate levels and the source's halo value at even ones."""
        from oracle import oracle as O

        apps = apps or self.apps_per_launch
        self.calls["step2"] += 1
        if end <= begin:
            return
        s = np.ascontiguousarray(src.numpy())
        cur = s
        for level in range(1, apps + 1):
            cur = O.step(self.shape, cur, self.w)  # zeros outside the local interior
            if (level % 2 == 0 or self.dirichlet) and level < apps:
                for d, k in enumerate(self.h):
                    for side in (slice(0, k), slice(-k, None)):
                        idx = (slice(None),) * d + (side,)
                        cur[idx] = s[idx]
        idx = (slice(self.h[0] + begin, self.h[0] + end),) + tuple(slice(k, -k) for k in self.h[1:])
        dst.numpy()[idx] = cur[idx]

    def wrap(self, buf):
        """Periodic halo of the local array (lora_plan_halo in wrap mode)."""
        x = buf.numpy()
        inner = tuple(slice(k, -k) for k in self.h)
        x[...] = np.pad(x[inner], [(k, k) for k in self.h], mode="wrap")

    def step_region(self, src, dst, begin, end):
        from oracle import oracle as O

        self.calls["step"] += 1
        if end <= begin:
            return
        h0 = self.h[0]
        sub = np.ascontiguousarray(src.numpy()[begin:end + 2 * h0])
        out = O.step(self.shape, sub, self.w)
        inner = tuple(slice(k, out.shape[i] - k) for i, k in enumerate(self.h))
        d = dst.numpy()[begin:end + 2 * h0]
        d[inner] = out[inner]


class OracleStepperBf16:
    """bf16 stand-in for HipStepper (tests only): tensors are torch.bfloat16, the oracle works on bit patterns."""

    wants_fused = True

    def __init__(self, layout, weights):
        self.shape = layout.shape
        self.w = np.ascontiguousarray(weights, dtype=np.float32)
        self.calls = {"step": 0, "step2": 0}

    def _sweep(self, s):
        import ctypes

        from oracle import oracle as O

        out = np.zeros_like(s)
        u16 = ctypes.POINTER(ctypes.c_uint16)
        fp = ctypes.POINTER(ctypes.c_float)
        sep = O.separable_27(self.w)  # the engine's default: exactly separable taps run as x/y/z passes
        if sep is not None:
            O.lib().oracle_step_3d_bf16_sep(s.ctypes.data_as(u16), out.ctypes.data_as(u16),
                                            *(f.ctypes.data_as(fp) for f in sep), *s.shape, 1)
        else:
            O.lib().oracle_step_3d_bf16(s.ctypes.data_as(u16), out.ctypes.data_as(u16), self.w.ctypes.data_as(fp),
                                        *s.shape, 1)
        return out

    def step2_region(self, src, dst, begin, end):
        """Two applications, level 1 = 0 outside the local interior (lora_plan_step2_region on bf16 grids)."""
        self.calls["step2"] += 1
        if end <= begin:
            return
        mid = self._sweep(np.ascontiguousarray(src.view(torch.int16).numpy().view(np.uint16)))  # halo stays 0
        out = self._sweep(mid)
        d = dst.view(torch.int16).numpy().view(np.uint16)
        d[1 + begin:1 + end, 2:-2, 4:-4] = out[1 + begin:1 + end, 2:-2, 4:-4]

    def step_region(self, src, dst, begin, end):
        self.calls["step"] += 1
        if end <= begin:
            return
        s = np.ascontiguousarray(src.view(torch.int16).numpy().view(np.uint16)[begin:end + 2])
        out = self._sweep(s)
        d = dst.view(torch.int16).numpy().view(np.uint16)[begin:end + 2]
        d[1:-1, 2:-2, 4:-4] = out[1:-1, 2:-2, 4:-4]


def _worker_bf16(rank, world, port, shape, dims, times, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lorastencil_amd import slab
        from oracle import oracle as O

        bits = O.to_bf16(O.reference_input(shape, dims))
        w = O.effective_weights(shape)
        drv = slab.SlabDriver(shape, dims, device="cpu", dtype="bf16", exchange_every=2,
                              stepper_factory=lambda lay: OracleStepperBf16(lay, w))
        drv.load_global(torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16))
        drv.run(times)
        full = drv.gather_global(0)
        if rank == 0:
            q.put(full.view(torch.int16).numpy().view(np.uint16))
    finally:
        dist.destroy_process_group()


def test_bf16_slabs_equal_single_rank(engine_built):
    from oracle import oracle as O

    shape, dims, times, world = "box3d1r", (12, 6, 16), 5, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, shape, dims, times, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bits = O.to_bf16(O.reference_input(shape, dims))
    assert np.array_equal(got, O.run_bf16(shape, bits, times))


def _worker(rank, world, port, shape, dims, times, overlap, q, fused=None, exchange_every=None, boundary="reference",
            apps=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lorastencil_amd import slab
        from oracle import oracle as O

        a = O.reference_input(shape, dims)
        w = O.effective_weights(shape)
        drv = slab.SlabDriver(shape, dims, device="cpu", overlap=overlap, boundary_rows=4, fused=fused,
                              exchange_every=exchange_every, boundary=boundary,
                              stepper_factory=lambda lay: OracleStepper(lay, w, wants_fused=fused is not False,
                                                                        dirichlet=boundary == "dirichlet", apps=apps))
        drv.load_global(a)
        # split the run in two calls: the driver must be resumable at any time level
        drv.run(times // 2)
        drv.run(times - times // 2)
        full = drv.gather_global(0)
        if rank == 0:
            q.put((full.numpy(), drv.fused, drv.exchange_every, drv.layout.ghost, dict(drv.stepper.calls)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_slabs(world, shape, dims, times, overlap=True, fused=None, exchange_every=None, info=False,
              boundary="reference", apps=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, dims, times, overlap, q, fused, exchange_every,
                                               boundary, apps))
             for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out if info else out[0]


@pytest.mark.parametrize("world,shape,dims,times", [
    (2, "star2d1r", (128, 64), 3),
    (2, "box2d3r", (64, 32), 4),
    (3, "star2d3r", (192, 32), 2),
    (2, "star3d1r", (12, 8, 16), 3),
    (3, "box3d1r", (9, 6, 8), 4),
    (2, "1d1r", (8192,), 3),
])
def test_slabs_equal_single_rank(engine_built, world, shape, dims, times):
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    expect = O.run(shape, a, times)
    got = run_slabs(world, shape, dims, times)
    if expect.ndim == 1:  # the host operator skips the last element; the slab driver returns the whole buffer
        expect[-1] = got[-1]
    assert got.shape == expect.shape
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("world,shape,dims,times,every", [
    (2, "star2d1r", (128, 64), 7, 1),     # fused pairs + odd tail, ghost zone refreshed after every launch
    (2, "star2d1r", (128, 64), 10, 4),    # communication-avoiding: 24-row ghost zones, exchange every 4 launches
    (3, "star2d3r", (192, 32), 9, 2),
    (2, "star3d1r", (16, 8, 16), 7, 3),   # fused pairs in 3D: 6-plane ghost zones exchanged every 3 launches
    (3, "box3d1r", (30, 6, 8), 9, 2),
    (2, "star3d1r", (16, 8, 16), 5, 3),
    (3, "box2d3r", (192, 32), 5, 2),      # 49-tap box: the stepper declines fusion
    (2, "1d2r", (8192,), 5, 2),           # fewer steps than one 8-step launch: single sweeps only
    (2, "1d1r", (8192,), 19, 2),          # 8-step fused launches in 1D: 64-point ghost zones, + 3 single sweeps
    (3, "1d2r", (12288,), 33, 1),
])
def test_ghost_zone_schedules_equal_single_rank(engine_built, world, shape, dims, times, every):
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    expect = O.run(shape, a, times)
    fused = None if shape != "box2d3r" and (shape, times) != ("star3d1r", 5) else False  # also 3D single sweeps
    got, was_fused, e, ghost, calls = run_slabs(world, shape, dims, times, fused=fused, exchange_every=every, info=True)
    if expect.ndim == 1:
        expect[-1] = got[-1]
    assert np.array_equal(got, expect)
    radius = {1: 4, 2: 3, 3: 1}[len(dims)]
    apps = (8 if len(dims) == 1 else 2) if was_fused else 1
    assert e == every and ghost == radius * apps * every
    assert was_fused == (fused is None)
    if was_fused and times // 2 >= apps:  # the run is split in two calls of times // 2 and the rest
        assert calls["step2"] > 0


@pytest.mark.parametrize("world,shape,dims,times,every", [
    (2, "star2d1r", (128, 64), 11, 1),    # four applications per launch (row-streaming kernel): 12-row ghost zones;
    (2, "star2d1r", (192, 64), 23, 2),    # the run is split 11 + 12: 4 + 4 + 2 + 1 and 4 + 4 + 4
    (3, "star2d3r", (192, 32), 14, 1),
])
def test_four_application_launches_across_slabs(engine_built, world, shape, dims, times, every):
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    expect = O.run(shape, a, times)  # 23 steps of the integer taps leave the exact range: compared to rounding below
    got, was_fused, e, ghost, calls = run_slabs(world, shape, dims, times, exchange_every=every, info=True, apps=4)
    assert was_fused and e == every and ghost == 3 * 4 * every
    assert calls["step2"] > 0 and (times != 23 or calls["two"] > 0)
    scale = np.abs(expect).max()
    assert np.abs(got - expect).max() <= 1e-12 * scale


@pytest.mark.parametrize("world,shape,dims,times,fused", [
    (2, "star2d1r", (128, 64), 7, None), (3, "box3d1r", (30, 6, 8), 6, None), (2, "1d1r", (8192,), 21, None),
    (2, "star2d3r", (128, 32), 5, False),
])
def test_dirichlet_slabs_equal_single_rank(engine_built, world, shape, dims, times, fused):
    """boundary="dirichlet": the caller's halo ring is kept at every time level, also inside fused launches."""
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    expect = O.run_bc(shape, a, times, "dirichlet")
    got = run_slabs(world, shape, dims, times, fused=fused, exchange_every=2, boundary="dirichlet")
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("world,shape,dims,times", [(3, "star2d1r", (192, 64), 9), (2, "star3d1r", (16, 8, 16), 7),
                                                    (2, "1d1r", (8192,), 19)])
def test_allgather_exchange_fallback_equals_single_rank(engine_built, monkeypatch, world, shape, dims, times):
    """LORA_SLAB_EXCHANGE=allgather: the ghost exchange through one all-gather of the boundary strips (what the driver
    falls back to when the backend refuses point-to-point operations)."""
    from oracle import oracle as O

    monkeypatch.setenv("LORA_SLAB_EXCHANGE", "allgather")
    a = O.reference_input(shape, dims)
    expect = O.run(shape, a, times)
    got = run_slabs(world, shape, dims, times, exchange_every=2)
    if expect.ndim == 1:
        expect[-1] = got[-1]
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("world,shape,dims,times,every", [
    (2, "star2d1r", (128, 64), 5, 2), (3, "star2d3r", (192, 32), 4, 1), (3, "box3d1r", (30, 6, 8), 6, 3),
    (2, "star3d1r", (16, 8, 16), 5, 2), (2, "1d1r", (8192,), 7, 2), (1, "star2d1r", (64, 64), 3, 1),
])
def test_periodic_slabs_equal_single_rank(engine_built, world, shape, dims, times, every):
    """boundary="periodic": the slabs form a ring, the unsplit dimensions wrap locally before every sweep."""
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    expect = O.run_bc(shape, a, times, "periodic")
    got = run_slabs(world, shape, dims, times, exchange_every=every, boundary="periodic")
    assert got.shape == expect.shape
    assert np.array_equal(got, expect)


def test_slabs_without_overlap_path(engine_built):
    from oracle import oracle as O

    shape, dims, times = "star2d1r", (64, 64), 2
    a = O.reference_input(shape, dims)
    assert np.array_equal(run_slabs(2, shape, dims, times, overlap=False), O.run(shape, a, times))


def test_slab_layout_properties(engine_built):
    from lorastencil_amd import slab

    for world in (1, 2, 4, 8):
        lays = [slab.slab_layout("star2d1r", (16384, 16384), world, r) for r in range(world)]
        assert lays[0].begin == 0 and lays[-1].end == 16384
        assert all(a.end == b.begin for a, b in zip(lays, lays[1:]))
        assert all((l.end - l.begin) % 32 == 0 for l in lays)
    lays = [slab.slab_layout("star3d1r", (510, 64, 64), 8, r) for r in range(8)]
    assert sum(l.end - l.begin for l in lays) == 510
    with pytest.raises(ValueError):
        slab.slab_layout("star2d1r", (64, 64), 8, 0)
