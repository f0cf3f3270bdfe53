"""The C++ multi-GPU slab driver (csrc/slab.cpp, group D of include/lorastencil.h).

CPU: the descriptor / error paths that need no device.  GPU: N slabs in one process on ONE device over the loopback
exchange (the complete driver: layout, ghost bookkeeping, boundary-first overlap, deferred wait, fused launches and
tails) against the single-GPU engine and the oracle, bit for bit; the RCCL backend through a ring of one slab; the CLI's
--gpus N."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from conftest import ROOT


@pytest.fixture(scope="module")
def L(engine_built):
    import lorastencil_amd as L

    return L


def has_gpu():
    import torch

    return torch.cuda.is_available()


def test_slab_descriptor_validation(L):
    from lorastencil_amd import _lib, cslab

    with pytest.raises(L.LoraError) as e:  # rank outside the decomposition
        cslab.SlabSet("star2d1r", (128, 64), nranks=2, ranks=[2], comms=(_lib.SlabComm * 1)())
    assert e.value.status == _lib.LORA_EINVAL
    with pytest.raises(L.LoraError) as e:  # two slabs need an exchange
        cslab.SlabSet("star2d1r", (128, 64), nranks=2, ranks=[0])
    assert e.value.status == _lib.LORA_EINVAL
    with pytest.raises(L.LoraError) as e:
        cslab.SlabSet("star2d1r", (128, 64), nranks=1, boundary="periodic")
    assert e.value.status == _lib.LORA_EUNSUPPORTED
    if not has_gpu():
        with pytest.raises(L.LoraError) as e:  # no CPU fallback here either
            cslab.SlabSet("star2d1r", (128, 64), nranks=1)
        assert e.value.status == _lib.LORA_ENODEVICE
        a = L.reference_input("star2d1r", (64, 64))
        with pytest.raises(L.LoraError) as e:
            cslab.run_host_multi("star2d1r", a, 2, times=1)
        assert e.value.status == _lib.LORA_ENODEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dims,nranks,times,every,flags,opts", [
    ("star2d1r", (384, 256), 3, 11, 0, 16, {"steps_per_launch": 4}),    # four applications per launch: 4 + 4 + 2 + 1; strips first
    ("star2d1r", (768, 384), 2, 23, 2, 0, {"steps_per_launch": 4}),     # 24-row ghost zones refreshed every 2 launches, deferred waits
    ("star2d1r", (768, 384), 4, 16, 1, 1, None),     # six per launch (the default: workgroup-row kernel); no boundary-first overlap
    ("star2d1r", (768, 384), 4, 16, 2, 2, None),     # no deferred wait
    ("star2d1r", (768, 384), 4, 22, 2, 16, None),    # boundary strips first (2D slabs go whole by default), deferred waits
    ("star2d1r", (768, 384), 3, 21, 1, 0, None),     # 6 + 6 + 6 + 2 + 1
    ("box2d3r", (300, 130), 3, 6, 3, 0, None),
    ("star2d3r", (256, 200), 2, 9, 1, 4, None),      # single sweeps only
    ("star3d1r", (24, 20, 64), 3, 7, 2, 0, None),
    ("star3d1r", (24, 20, 64), 3, 9, 1, 16, None),   # boundary planes first (3D slabs go whole by default too)
    ("box3d1r", (30, 16, 64), 2, 8, 0, 0, None),
    ("star3d1r", (60, 40, 130), 3, 11, 1, 0, {"steps_per_launch": 4}),  # the register-resident kernel in z-slabs: 4 + 4 + 2 + 1
    ("box3d1r", (64, 30, 250), 4, 14, 2, 16, {"steps_per_launch": 4}),
    ("1d1r", (30000,), 3, 27, 2, 0, None),           # eight applications per launch + single sweeps
])
def test_loopback_slabs_equal_single_gpu(L, shape, dims, nranks, times, every, flags, opts):
    from lorastencil_amd import cslab
    from oracle import oracle as O

    a = O.reference_input(shape, dims)
    # one slab, same launch schedule (the run is split in two calls below): N slabs must equal it BIT FOR BIT -- same
    # kernels, same per-point arithmetic whatever the decomposition -- even where values have left the exact range
    one = cslab.SlabSet(shape, dims, 1, flags=flags & 4, options=opts)
    one.load(a)
    one.run(times // 2)
    one.run(times - times // 2)
    single = one.store(np.zeros_like(a))
    slabs = cslab.SlabSet(shape, dims, nranks, comms=cslab.loopback_comms(nranks), exchange_every=every, flags=flags,
                          options=opts)
    si = slabs.info(0)
    assert si.apps_per_launch == (1 if flags & 4 else (opts or {}).get("steps_per_launch", {1: 8, 2: 6, 3: 2}[len(dims)]))
    assert si.ghost == {1: 4, 2: 3, 3: 1}[len(dims)] * si.apps_per_launch * si.exchange_every
    slabs.load(a)
    slabs.run(times // 2)          # resumable at any time level
    slabs.run(times - times // 2)
    out = np.zeros_like(a)
    slabs.store(out)
    assert np.array_equal(out, single)   # N slabs == 1 slab bit for bit, the halo state included
    exp = O.run(shape, a, times)
    if a.ndim == 1:
        exp[-1] = out[-1]
    if np.abs(exp).max() < 2.0 ** 50:
        assert np.array_equal(out, exp)
    else:
        assert np.abs(out - exp).max() <= 1e-13 * np.abs(exp).max()
    si = slabs.info(0)
    assert si.steps_done == times and (si.exchanges > 0 or times <= si.apps_per_launch * si.exchange_every)


@pytest.mark.gpu
def test_loopback_slabs_dirichlet_and_bf16(L):
    from lorastencil_amd import cslab
    from oracle import oracle as O

    shape, dims = "star2d1r", (384, 256)
    a = O.reference_input(shape, dims)
    w = O.effective_weights(shape)
    w = w / w.sum()
    slabs = cslab.SlabSet(shape, dims, 3, comms=cslab.loopback_comms(3), boundary="dirichlet", weights=w)
    assert slabs.info(0).apps_per_launch == 4   # the workgroup-row kernel with a row of halo values per level (six: reference boundary only)
    slabs.load(a)
    slabs.run(9)
    out = slabs.store(np.zeros_like(a))
    exp = O.run_bc(shape, a, 9, "dirichlet", weights=w)
    assert np.abs(out - exp).max() <= 1e-13 * np.abs(exp).max()
    # bf16 planes
    shape, dims = "box3d1r", (24, 20, 64)
    bits = O.to_bf16(O.reference_input(shape, dims))
    slabs = cslab.SlabSet(shape, dims, 3, comms=cslab.loopback_comms(3), dtype="bf16")
    slabs.load(bits)
    slabs.run(6)
    out = slabs.store(np.zeros_like(bits))
    assert np.array_equal(out, O.run_bf16(shape, bits, 6))
    # bf16 z-slabs through the register-resident kernel: four per launch (4 + 4 + 2 + 1), ghost zones of 4 x E planes
    dims = (64, 30, 248)
    rng = np.random.default_rng(3)
    bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
    w = O.effective_weights(shape)
    w = w / w.sum()
    for nranks, every, flags in ((4, 2, 16), (3, 1, 0)):
        slabs = cslab.SlabSet(shape, dims, nranks, comms=cslab.loopback_comms(nranks), dtype="bf16", weights=w, exchange_every=every,
                              flags=flags, options={"steps_per_launch": 4})
        assert slabs.info(0).apps_per_launch == 4 and slabs.info(0).ghost == 4 * every
        slabs.load(bits)
        slabs.run(5)
        slabs.run(6)
        out = slabs.store(np.zeros_like(bits))
        assert np.array_equal(out, O.run_bf16(shape, bits, 11, weights=w)), (nranks, every, flags)


@pytest.mark.gpu
def test_rccl_backend_through_the_c_driver_ring_of_one(L):
    """ncclSend / ncclRecv inside ncclGroupStart / End on the slab's communication stream: a single slab closed into a
    ring sends its boundary strips to itself through RCCL (two RCCL ranks need two GPUs).  Under the reference boundary
    a ring of one is a rehearsal, not a physical result -- what must hold is that the overlapped, deferred schedule
    over RCCL equals the plain schedule over the loopback copies bit for bit."""
    from lorastencil_amd import cslab
    from oracle import oracle as O

    rccl = ctypes.CDLL("librccl.so.1")
    comm = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, dev) == 0
    try:
        for shape, dims, times in (("star2d1r", (512, 384), 23), ("star3d1r", (48, 20, 64), 9), ("1d1r", (30000,), 19)):
            a = O.reference_input(shape, dims)
            res = []
            for comms, flags in (((cslab._lib.SlabComm * 1)(cslab.rccl_comm(comm.value)), cslab.SLAB_RING_OF_ONE),
                                 (cslab.loopback_comms(1), cslab.SLAB_RING_OF_ONE | cslab.SLAB_NO_OVERLAP | cslab.SLAB_NO_DEFER)):
                slabs = cslab.SlabSet(shape, dims, 1, comms=comms, exchange_every=2, flags=flags)
                assert slabs.info(0).ghost_top == slabs.info(0).ghost > 0
                slabs.load(a)
                slabs.refresh_ghosts()
                slabs.run(times)
                slabs.sync()
                res.append(slabs.store(np.zeros_like(a)))
                assert slabs.info(0).exchanges >= 2
                slabs.close()
            assert np.array_equal(res[0], res[1]), shape
    finally:
        rccl.ncclCommDestroy(comm)


@pytest.mark.gpu
def test_cli_gpus_flag_over_loopback(L):
    """`lorastencil_2d star2d1r 384 256 9 --gpus=3`: the reference's surface with N slabs (LORA_SLAB_LOOPBACK=1: three
    slabs on this one GPU); same three stdout lines, same result as one GPU."""
    from lorastencil_amd import cslab
    from oracle import oracle as O

    exe = os.path.join(ROOT, "lorastencil_amd", "bin", "lorastencil_2d")
    env = dict(os.environ, LORA_SLAB_LOOPBACK="1")
    p = subprocess.run([exe, "star2d1r", "384", "256", "9", "--gpus=3"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.splitlines()
    assert lines[0].startswith("INFO: shape = star_2d1r") and "LoRAStencil(2D star_2d1r): " in lines
    assert any(l.startswith("GStencil/s = ") for l in lines) and "GPUs = 3 (row / plane slabs, RCCL ghost-zone exchange)" in lines
    p = subprocess.run([exe, "star2d1r", "384", "256", "9", "--gpus=64"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "more GPUs requested" in p.stdout
    os.environ["LORA_SLAB_LOOPBACK"] = "1"
    try:
        a = O.reference_input("star2d1r", (384, 256))
        out, info = cslab.run_host_multi("star2d1r", a, 3, times=9)
    finally:
        del os.environ["LORA_SLAB_LOOPBACK"]
    exp = O.run("star2d1r", a, 9)  # 100^9 has left the exact-integer range: compared to rounding
    assert np.abs(out - exp).max() <= 1e-13 * np.abs(exp).max() and info.steps_per_launch == 6
    assert np.array_equal(out[:4], exp[:4]) and np.array_equal(out[:, :4], exp[:, :4])  # halo state after an odd run: 0
