"""The C++ two-axis block driver (csrc/blocks.cpp, include/lorastencil.h group E) against the oracle: Pa x Pb blocks in one
process on one device (loopback exchange) == the undivided grid, bit for bit -- 2D shapes (rows x columns) and 3D shapes
(planes x rows), fp64 and bf16, fused launches with their tails, every refresh interval, resumed runs."""
import os
import subprocess

import numpy as np
import pytest
from conftest import ROOT, has_gpu


@pytest.fixture(scope="module")
def L(engine_built):
    import lorastencil_amd as L

    return L


def test_block_driver_argument_checks(L):
    from lorastencil_amd import _lib, cblocks

    with pytest.raises(L.LoraError) as e:
        cblocks.Block("1d1r", (4096,), (1, 1), (0, 0))  # 1D grids have one outer dimension: slabs
    assert e.value.status == _lib.LORA_EUNSUPPORTED
    with pytest.raises(L.LoraError) as e:
        cblocks.Block("star2d1r", (128, 128), (2, 2), (2, 0))  # coordinates outside the grid
    assert e.value.status == _lib.LORA_EINVAL
    with pytest.raises(L.LoraError) as e:
        cblocks.Block("star2d1r", (128, 128), (2, 2), (0, 0))  # a cut grid needs a callback table
    assert e.value.status == _lib.LORA_EINVAL
    if not has_gpu():
        with pytest.raises(L.LoraError) as e:
            cblocks.Block("star2d1r", (128, 128), (1, 1), (0, 0))
        assert e.value.status == _lib.LORA_ENODEVICE  # no CPU fallback here either


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dims,grid,times,every,flags,opts", [
    ("star2d1r", (768, 1024), (2, 2), 16, 2, 0, None),      # six per launch (6 + 6 + 4 as 2 + 2): deferred wait
    ("star2d1r", (768, 1024), (2, 4), 21, 1, 0, None),      # the 2 x 4 grid of SURVEY 8 f4; 6 + 6 + 6 + 2 + 1
    ("star2d1r", (900, 700), (3, 2), 13, 2, 2, None),       # ragged extents; wait at once
    ("box2d3r", (512, 600), (2, 2), 9, 2, 0, {"steps_per_launch": 4}),
    ("star2d3r", (400, 512), (1, 3), 8, 1, 0, None),        # columns only
    ("star2d1r", (640, 256), (4, 1), 14, 2, 0, None),       # rows only: the slabs' case
    ("star2d1r", (512, 512), (2, 2), 7, 3, 4, None),        # single sweeps only
    ("star3d1r", (48, 40, 128), (2, 2), 9, 2, 0, None),     # planes x rows, two per launch (small grid)
    ("star3d1r", (96, 80, 136), (2, 3), 11, 1, 0, {"steps_per_launch": 4}),   # the register-resident kernel: 4 + 4 + 2 + 1
    ("box3d1r", (64, 96, 120), (4, 2), 13, 2, 0, {"steps_per_launch": 4}),
    ("box3d1r", (40, 30, 64), (1, 2), 6, 1, 0, None),       # rows only
])
def test_loopback_blocks_equal_the_undivided_grid(L, shape, dims, grid, times, every, flags, opts):
    from lorastencil_amd import cblocks
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    w = None
    if times > 6:  # keep the values in the exact range of fp64
        w = O.effective_weights(shape)
        w = w / w.sum()
        a = np.random.default_rng(2).standard_normal(O.padded_shape(shape, dims))
    n = grid[0] * grid[1]
    g = cblocks.BlockGrid(shape, dims, grid, comms=cblocks.loopback_comms(n), weights=w, exchange_every=every, flags=flags, options=opts)
    i0 = g.info(0)
    radius = 1 if len(dims) == 3 else 3
    assert i0.ghost == radius * i0.apps_per_launch * i0.exchange_every
    if opts and "steps_per_launch" in opts:
        assert i0.apps_per_launch == opts["steps_per_launch"]
    g.load(a)
    g.run(times // 2)  # resumable at any time level
    g.run(times - times // 2)
    out = g.store(np.zeros_like(a))
    exp = O.run(shape, a, times, weights=w)
    if w is None:
        assert np.array_equal(out, exp)
    else:
        # the same kernels on other extents: bit-identical per point for the direct-tap forms, to rounding for the
        # structured evaluations (their summation order does not depend on the extents either, but the oracle's differs)
        assert np.abs(out - exp).max() <= 1e-13 * np.abs(exp).max()
        one = cblocks.BlockGrid(shape, dims, (1, 1), weights=w, options=opts, flags=flags & 4)
        one.load(a)
        one.run(times // 2)
        one.run(times - times // 2)
        single = one.store(np.zeros_like(a))
        assert np.array_equal(out, single)  # N blocks == 1 block bit for bit, the halo state included
        one.close()
    assert g.info(0).steps_done == times and g.info(0).exchanges > 0
    g.close()


@pytest.mark.gpu
def test_loopback_blocks_bf16(L):
    from lorastencil_amd import cblocks
    from oracle import oracle as O

    shape, dims = "box3d1r", (64, 60, 248)
    bits = O.to_bf16(np.random.default_rng(5).standard_normal(O.padded_shape(shape, dims)))
    w = O.effective_weights(shape)
    w = w / w.sum()
    for grid, every, opts in (((2, 2), 2, None), ((2, 3), 1, {"steps_per_launch": 4}), ((4, 1), 2, {"steps_per_launch": 4})):
        n = grid[0] * grid[1]
        g = cblocks.BlockGrid(shape, dims, grid, comms=cblocks.loopback_comms(n), dtype="bf16", weights=w, exchange_every=every, options=opts)
        g.load(bits)
        g.run(5)
        g.run(6)
        out = g.store(np.zeros_like(bits))
        assert np.array_equal(out, O.run_bf16(shape, bits, 11, weights=w)), (grid, every, opts)
        g.close()


@pytest.mark.gpu
def test_cli_grid_flag_over_loopback(L):
    """`lorastencil_3d star3d1r 48 40 128 6 --grid=2x2` / `lorastencil_2d ... --grid=2x3`: the reference's surface on a grid of
    blocks (LORA_SLAB_LOOPBACK=1: all blocks on this one GPU); the same stdout lines, the same result as one GPU."""
    from lorastencil_amd import cblocks
    from oracle import oracle as O

    env = dict(os.environ, LORA_SLAB_LOOPBACK="1")
    for dim, argv, label in ((3, ["star3d1r", "48", "40", "128", "6", "--grid=2x2"], "GPUs = 4 (2 x 2 blocks of the two outer dimensions, RCCL ghost-zone exchange)"),
                             (2, ["star2d1r", "384", "512", "7", "--grid=2x3"], "GPUs = 6 (2 x 3 blocks of the two outer dimensions, RCCL ghost-zone exchange)")):
        exe = os.path.join(ROOT, "lorastencil_amd", "bin", f"lorastencil_{dim}d")
        p = subprocess.run([exe, *argv], capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        lines = p.stdout.splitlines()
        assert any(l.startswith("GStencil/s = ") for l in lines) and label in lines, p.stdout
    p = subprocess.run([exe, "star2d1r", "384", "512", "7", "--grid=8x9"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "more GPUs requested" in p.stdout
    os.environ["LORA_SLAB_LOOPBACK"] = "1"
    try:
        a = O.reference_input("star3d1r", (48, 40, 128))
        out, info = cblocks.run_host_blocks("star3d1r", a, (2, 2), times=6)
        assert np.array_equal(out, O.run("star3d1r", a, 6))
        bits = O.to_bf16(O.reference_input("box3d1r", (24, 20, 64)))
        out16, _ = cblocks.run_host_blocks("box3d1r", bits, (2, 2), times=4, dtype="bf16")
        assert np.array_equal(out16, O.run_bf16("box3d1r", bits, 4))
    finally:
        del os.environ["LORA_SLAB_LOOPBACK"]
