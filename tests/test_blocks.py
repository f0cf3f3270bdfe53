"""2-D block decomposition of the 2D shapes (lorastencil_amd/blocks.py).

CPU: the driver's layout, ghost bookkeeping, two-phase exchange (loopback and gloo P2P, one block per rank) and launch
schedule with an oracle-backed stepper standing in for the HIP engine.  GPU: every block of a Py x Px grid on one device,
the HIP kernels on each block and the pack / unpack kernel between them, against the undivided grid bit for bit."""
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


class OracleBlockStepper:
    """CPU stand-in for HipBlockStepper (tests only): the whole local grid per launch, cells outside the local interior 0
    at odd levels and the source's pad at even ones (lora_plan_stepk)."""

    def __init__(self, shape, layout, apps):
        from oracle import oracle as O

        self.shape, self.apps_per_launch, self.w = shape, apps, O.effective_weights(shape)

    def _run(self, src, dst, apps):
        from oracle import oracle as O

        s = np.ascontiguousarray(src.numpy())
        cur = s
        for level in range(1, apps + 1):
            cur = O.step(self.shape, cur, self.w)
            if level % 2 == 0 and level < apps:
                for idx in ((slice(0, 4),), (slice(-4, None),), (slice(None), slice(0, 4)), (slice(None), slice(-4, None))):
                    cur[idx] = s[idx]
        dst.numpy()[4:-4, 4:-4] = cur[4:-4, 4:-4]

    def step(self, src, dst):
        self._run(src, dst, 1)

    def step2(self, src, dst):
        self._run(src, dst, 2)

    def stepk(self, src, dst):
        self._run(src, dst, self.apps_per_launch)

    def copy_block(self, dst, dst_off, dst_ld, src, src_off, src_ld, rows, cols):
        d = dst.numpy().reshape(-1)
        s = src.numpy().reshape(-1)
        for r in range(rows):
            d[dst_off + r * dst_ld:dst_off + r * dst_ld + cols] = s[src_off + r * src_ld:src_off + r * src_ld + cols]


def _expected(shape, a, times):
    from oracle import oracle as O

    return O.run(shape, a, times)


def _close(out, exp):
    if np.abs(exp).max() < 2.0 ** 50:
        return np.array_equal(out, exp)
    return np.abs(out - exp).max() <= 1e-13 * np.abs(exp).max()


def test_block_layout_covers_the_grid_once():
    from lorastencil_amd.blocks import BlockLayout

    m, n, grid, g = 301, 130, (3, 4), 12
    seen = np.zeros((m, n), dtype=int)
    for iy in range(grid[0]):
        for ix in range(grid[1]):
            lay = BlockLayout((m, n), grid, (iy, ix), g)
            seen[lay.r0:lay.r1, lay.c0:lay.c1] += 1
            assert lay.r0 % 2 == 0 and lay.c0 % 2 == 0
            assert lay.local_dims == (lay.own[0] + (g if iy else 0) + (g if iy < 2 else 0),
                                      lay.own[1] + (g if ix else 0) + (g if ix < 3 else 0))
    assert (seen == 1).all()


@pytest.mark.parametrize("shape,dims,grid,apps,every,times", [
    ("star2d1r", (96, 80), (2, 2), 2, 2, 7),
    ("star2d1r", (96, 120), (2, 4), 4, 1, 11),    # 4 + 4 + 2 + 1
    ("box2d3r", (90, 64), (3, 1), 2, 2, 5),
    ("star2d3r", (60, 140), (1, 3), 6, 1, 15),    # 6 + 6 + 2 + 1
    ("box2d1r", (64, 64), (2, 2), 1, 3, 5),       # single sweeps: ring 0 at odd levels
])
def test_loopback_blocks_equal_the_undivided_grid_cpu(shape, dims, grid, apps, every, times):
    from lorastencil_amd import blocks
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    bs = blocks.BlockSet(shape, dims, grid, device="cpu", exchange_every=every,
                         stepper_factory=lambda lay: OracleBlockStepper(shape, lay, apps))
    assert bs.apps == apps and bs.ghost == 3 * apps * bs.exchange_every
    bs.load(a)
    bs.run(times // 2)
    bs.run(times - times // 2)   # resumable; an odd first half leaves the schedule on single sweeps until the level is even
    out = bs.store()
    assert _close(out, _expected(shape, a, times))
    assert bs.steps_done == times and bs.exchanges > 0


def test_blocks_reject_what_they_do_not_do():
    from lorastencil_amd import blocks

    with pytest.raises(ValueError):
        blocks.BlockSet("star3d1r", (16, 16, 64), (2, 2), device="cpu")
    with pytest.raises(ValueError):  # blocks thinner than the ghost zone
        blocks.BlockSet("star2d1r", (32, 32), (4, 4), device="cpu",
                        stepper_factory=lambda lay: OracleBlockStepper("star2d1r", lay, 6))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, dims, grid, apps, every, times, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lorastencil_amd import blocks
        from oracle import oracle as O

        a = O.reference_input(shape, dims)
        bs = blocks.BlockSet(shape, dims, grid, device="cpu", exchange_every=every, distributed=True,
                             stepper_factory=lambda lay: OracleBlockStepper(shape, lay, apps))
        assert len(bs.blocks) == 1
        bs.load(a)
        bs.run(times)
        out = bs.store()
        if rank == 0:
            q.put((out, bs.exchanges))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,dims,grid,apps,every,times", [
    ("star2d1r", (96, 80), (2, 2), 4, 1, 11),
    ("box2d3r", (64, 150), (1, 3), 2, 2, 7),
])
def test_one_block_per_rank_over_gloo(shape, dims, grid, apps, every, times):
    from oracle import oracle as O

    world = grid[0] * grid[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, dims, grid, apps, every, times, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, exchanges = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a = O.reference_input(shape, dims)
    assert _close(out, _expected(shape, a, times)) and exchanges > 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dims,grid,opts,every,times", [
    ("star2d1r", (768, 1536), (2, 4), None, 1, 21),                     # the 2 x 4 grid of SURVEY 8f-4; 6 + 6 + 6 + 2 + 1
    ("star2d1r", (640, 700), (2, 2), {"steps_per_launch": 4}, 2, 11),
    ("box2d3r", (300, 260), (3, 2), None, 1, 9),
    ("star2d3r", (512, 512), (1, 2), {"steps_per_launch": 2}, 2, 8),
    ("box2d1r", (256, 384), (2, 2), {"steps_per_launch": 1}, 3, 5),
])
def test_loopback_blocks_equal_single_gpu(engine_built, shape, dims, grid, opts, every, times):
    import lorastencil_amd as L
    from lorastencil_amd import blocks
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    bs = blocks.BlockSet(shape, dims, grid, options=opts, exchange_every=every)
    assert bs.apps == (opts or {}).get("steps_per_launch", 6)
    bs.load(a)
    bs.run(times // 2)
    bs.run(times - times // 2)
    out = bs.store()
    exp = _expected(shape, a, times)
    assert _close(out, exp)
    assert bs.exchanges > 0
    single, _ = L.run_host(shape, a, times=times)  # the single-GPU engine on the undivided grid
    assert _close(out[4:-4, 4:-4], single[4:-4, 4:-4])


def _gpu_worker(rank, world, port, shape, dims, grid, every, times, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lorastencil_amd import blocks
        from oracle import oracle as O

        a = O.reference_input(shape, dims)
        bs = blocks.BlockSet(shape, dims, grid, device="cuda:0", exchange_every=every, distributed=True)
        bs.load(a)
        bs.run(times // 2)
        bs.run(times - times // 2)
        out = bs.store()
        if rank == 0:
            q.put((out, bs.apps, bs.exchanges))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_four_ranks_one_block_each_on_one_gpu(engine_built):
    """The one-block-per-rank path with the real HIP stepper and pack / unpack kernel: four ranks share this GPU and
    exchange over gloo (RCCL needs one GPU per rank), a 2 x 2 grid against the undivided grid."""
    from oracle import oracle as O

    shape, dims, grid, every, times = "star2d1r", (640, 768), (2, 2), 2, 15
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 4, port, shape, dims, grid, every, times, q)) for r in range(4)]
    for p in procs:
        p.start()
    out, apps, exchanges = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    a = O.reference_input(shape, dims)
    assert apps == 6 and exchanges > 0
    assert _close(out, _expected(shape, a, times))
