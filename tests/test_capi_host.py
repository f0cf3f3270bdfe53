"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/lorastencil.h declares, its
host helpers agree with the oracle, argument validation works, and compute entry points fail loudly without a
GPU (no fallback)."""
import os
import re

import numpy as np
import pytest
from conftest import ALL_SHAPES, ROOT, has_gpu, load_golden

from oracle import oracle as O


@pytest.fixture(scope="module")
def L(engine_built):
    import lorastencil_amd as L

    return L


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lorastencil.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lora_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(L):
    import ctypes

    from lorastencil_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lorastencil.h but not exported"
    # and the Python binding table covers the same set
    assert sorted(_lib.SIGNATURES) == names


def test_shim_library_exports_reference_signatures(engine_built):
    import subprocess

    path = os.path.join(ROOT, "lorastencil_amd", "lib", "liblorastencil_shims.so")
    syms = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    # Itanium manglings of the reference prototypes (1d_utils.h:45-47, 2d_utils.h:47-51, 3d_utils.h:44-48)
    for m in ["_Z8gpu_1d1rPKdPdS0_ii", "_Z8gpu_1d2rPKdPdS0_ii", "_Z13gpu_star_2d1rPKdPdS0_iii",
              "_Z13gpu_star_2d3rPKdPdS0_iii", "_Z12gpu_box_2d3rPKdPdS0_iii", "_Z12gpu_box_3d1rPKdPdS0_iiii",
              "_Z13gpu_star_3d1rPKdPdS0_iiii"]:
        assert m in syms, m


@pytest.mark.parametrize("shape", ALL_SHAPES)
def test_params_and_weights_match_oracle(L, shape):
    assert np.array_equal(L.default_params(shape), O.default_params(shape))
    assert np.array_equal(L.effective_weights(shape), O.effective_weights(shape))
    rng = np.random.default_rng(7)
    p = rng.standard_normal(O.NTAPS[O.NDIM[O.SHAPES[shape]]])
    if shape.startswith("box2d"):
        p = p.reshape(7, 7)
        p = p + p.T + p[::-1] + p[:, ::-1]  # the pyramid factoriser assumes a symmetric matrix
        p = (p + p[::-1, ::-1]).ravel()
    assert np.array_equal(L.effective_weights(shape, p), O.effective_weights(shape, p))


def test_factorizer_matches_oracle_and_reports_residual(L):
    p = L.default_params("box2d3r")
    u, v, res = L.factorize_7x7(p)
    uo, vo = O.factorize_7x7(p)
    assert np.array_equal(u, uo) and np.array_equal(v, vo) and res == 0.0
    p2 = p.copy()
    p2[24] = 10.0
    u, v, res = L.factorize_7x7(p2)
    uo, vo = O.factorize_7x7(p2)
    assert np.array_equal(u, uo) and np.array_equal(v, vo) and res == 2.0
    # reconstruction of the first three terms = what the operator applies
    w = sum(np.outer(u[t], v[t]) for t in range(3))
    assert np.array_equal(w.ravel(), L.effective_weights("box2d3r", p2))


def test_svd_factorisation_reveals_rank_and_bounds_the_error(L):
    rng = np.random.default_rng(21)
    # exactly rank 2, not symmetric: the pyramid scheme cannot take it, the SVD needs two terms
    w = np.outer(rng.standard_normal(7), rng.standard_normal(7)) + np.outer(rng.standard_normal(7), rng.standard_normal(7))
    u, v, sig = L.svd_7x7(w)
    assert np.all(np.diff(sig) <= 1e-12) and sig[2] < 1e-12 * sig[0]
    assert np.abs(sum(np.outer(u[k], v[k]) for k in range(2)) - w).max() < 1e-13 * np.abs(w).max()
    # full-rank random taps: every truncation obeys Eckart-Young and agrees with numpy's singular values
    w = rng.standard_normal((7, 7))
    u, v, sig = L.svd_7x7(w)
    assert np.allclose(sig, np.linalg.svd(w, compute_uv=False), rtol=1e-12, atol=1e-13)
    for r in range(1, 8):
        resid = w - sum(np.outer(u[k], v[k]) for k in range(r))
        bound = sig[r] if r < 7 else 0.0
        assert np.linalg.norm(resid, 2) <= bound * (1 + 1e-9) + 1e-12
    # the reference's box table really is rank 3, its star2d1r table rank 4
    assert (L.svd_7x7(L.default_params("box2d3r"))[2] > 1e-9).sum() == 3
    assert (L.svd_7x7(L.default_params("star2d1r"))[2] > 1e-9).sum() == 4
    p = L.Plan("box2d3r", (64, 128))
    p.set_weights(np.outer([1, 2, 3, 4, 5, 6, 7], [1, -1, 2, -2, 3, -3, 4.0]))  # rank 1, asymmetric
    p.set_variant(L.VARIANT_MFMA)  # accepted through the SVD path
    p.set_weights(rng.standard_normal(49))  # rank 7: no low-rank form, the plan falls back to the direct variant
    assert p.get_option("variant") == L.VARIANT_DIRECT


def test_glibc_rand_fill_matches_oracle_and_golden(L):
    r = L.GlibcRand()
    ro = O.Rng()
    assert [r.next() for _ in range(1000)] == [ro.next() for _ in range(1000)]
    for shape in ("1d1r", "star2d1r", "box3d1r"):
        g = load_golden(shape)
        assert np.array_equal(L.reference_input(shape, tuple(g["dims"])), g["input"])


def test_separable_tap_test_matches_oracle(L):
    """bf16 plans evaluate exactly separable 3x3x3 taps as x/y/z passes; engine and oracle must take the same
    decision and read off the same fp32 factors (a different decision would change the summation order)."""
    rng = np.random.default_rng(29)
    cases = [O.effective_weights("box3d1r"), O.effective_weights("box3d1r") / 36.0, O.effective_weights("star3d1r"),
             np.zeros(27), np.einsum("k,i,j->kij", [1, 2, 1], [0.5, 1, 0.5], [3, 5, 7.0]).ravel()]
    for _ in range(50):  # every box3d1r of the reference's API: w[dz][dy][dx] = params[dx]
        prm = np.zeros(27)
        prm[:3] = rng.standard_normal(3)
        cases.append(O.effective_weights("box3d1r", prm))
    for _ in range(50):  # rank-1 in fp64; rarely exactly so after the fp32 cast
        a, b, c = rng.standard_normal((3, 3))
        cases.append(np.einsum("k,i,j->kij", a, b, c).ravel())
    cases.append(rng.standard_normal(27))
    n_sep = 0
    for w in cases:
        e, o = L.separable_3x3x3(w), O.separable_27(w)
        assert (e is None) == (o is None)
        if e is not None:
            n_sep += 1
            for fe, fo in zip(e, o):
                assert fe.dtype == np.float32 and np.array_equal(fe, fo)
            c, b, a = (f.astype(np.float32) for f in e)
            rebuilt = ((a[:, None, None] * b[None, :, None]).astype(np.float32) * c[None, None, :]).astype(np.float32)
            assert np.array_equal(rebuilt.ravel(), np.asarray(w, dtype=np.float64).astype(np.float32))
    assert n_sep >= 53  # the two reference tables, the anisotropic case and all 50 params[0..2] draws
    assert L.separable_3x3x3(O.effective_weights("star3d1r")) is None
    plan = L.Plan("box3d1r", (8, 8, 16), dtype="bf16")
    assert plan.get_option("tapset") == 2 and plan.get_option("separable") == -1
    assert plan.set_option("separable", 0).get_option("tapset") == 1
    assert L.Plan("box3d1r", (8, 8, 16)).get_option("tapset") == 1  # fp64 keeps the 27-tap order (HBM-bound)


def test_bf16_conversion_matches_oracle(L):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.uniform(-35, 35, 20000),
                        [0.0, -0.0, np.inf, -np.inf, 257.0, 259.0, 3.3e38, 3.4e38, 1e-40]])
    x = x.astype(np.float32).astype(np.float64)
    assert np.array_equal(L.to_bf16(x), O.to_bf16(x))
    b = L.to_bf16(x)
    assert np.array_equal(L.from_bf16(b), O.from_bf16(b))
    assert L.from_bf16(L.to_bf16(np.array([257.0, 259.0]))).tolist() == [256.0, 260.0]  # ties to even
    with pytest.raises(L.LoraError):
        L.Plan("star2d1r", (64, 128), dtype="bf16")  # bf16 is a 3D-only path
    with pytest.raises(L.LoraError):
        L.Plan("box3d1r", (8, 8, 12), dtype="bf16")  # innermost extent must be a multiple of 8
    assert L.Plan("box3d1r", (8, 8, 16), dtype="bf16").kernel_name == "stencil3d_bf16_fused2_kernel"
    assert L.Plan("box3d1r", (8, 8, 16), dtype="bf16").set_option("steps_per_launch", 1).kernel_name == \
        "stencil3d_bf16_kernel"


def test_shape_tables(L):
    from lorastencil_amd import _lib

    lib = _lib.lib()
    info = [lib.lora_shape_info_name(s).decode() for s in range(8)]
    assert info == ["1d1r", "1d2r", "star_2d1r", "box_2d1r", "star_2d3r", "box_2d3r", "star_3d1r", "box_3d1r"]
    assert [lib.lora_shape_gstencil_factor(s) for s in range(8)] == [3, 2, 3, 3, 1, 3, 1, 1]
    assert lib.lora_shape_from_name(b"star2d1r") == 2 and lib.lora_shape_from_name(b"nope") < 0
    assert L.padded_shape("star3d1r", (512, 512, 512)) == (514, 516, 520)


def test_plan_validation(L):
    # odd innermost extent: the fused 2D kernels (six applications: workgroup rows; four / two: row streaming) take it,
    # the generic fallback serves single sweeps / 3D
    assert L.Plan("star2d1r", (64, 127)).kernel_name == "stencil2d_wg_kernel"
    assert L.Plan("star2d1r", (64, 127)).set_option("steps_per_launch", 4).kernel_name == "stencil2d_stream_kernel"
    assert L.Plan("star2d1r", (64, 127)).set_option("steps_per_launch", 1).kernel_name == "stencil2d_generic_kernel"
    assert L.Plan("box3d1r", (4, 4, 5)).kernel_name == "stencil3d_generic_kernel"
    with pytest.raises(L.LoraError):
        L.Plan("star2d1r", (0, 128))
    p = L.Plan("star2d1r", (64, 128))
    assert p.get_option("tapset") == 0  # 25-tap diamond
    assert L.Plan("star2d3r", (64, 128)).get_option("tapset") == 1
    assert L.Plan("box2d3r", (64, 128)).get_option("tapset") == 2
    assert L.Plan("star3d1r", (8, 16, 128)).get_option("tapset") == 0
    assert L.Plan("box3d1r", (8, 16, 128)).get_option("tapset") == 1
    p.set_weights(np.ones(49))
    assert p.get_option("tapset") == 2
    with pytest.raises(L.LoraError):
        p.set_weights(np.ones(9))
    with pytest.raises(L.LoraError):
        p.set_option("rows_per_thread", 5)
    with pytest.raises(L.LoraError):
        p.step(0, 0)  # null buffers


def test_block_copy_argument_checks(L):
    """lora_copy_block_f64 (the pack / unpack kernel of the 2-D block decomposition): what it refuses before any launch"""
    from lorastencil_amd import _lib

    f = _lib.lib().lora_copy_block_f64
    assert f(None, 8, 4096, 8, 2, 2, None) == _lib.LORA_EINVAL          # null destination
    assert f(4096, 8, None, 8, 2, 2, None) == _lib.LORA_EINVAL          # null source
    assert f(4096, 8, 8192, 8, -1, 2, None) == _lib.LORA_EINVAL         # negative extent
    assert f(4096, 3, 8192, 8, 2, 4, None) == _lib.LORA_EINVAL          # rows overlap: leading dimension < columns
    assert f(4100, 8, 8192, 8, 2, 2, None) == _lib.LORA_EUNSUPPORTED    # not 8-byte aligned
    assert f(4096, 8, 8192, 8, 0, 0, None) == _lib.LORA_OK              # nothing to copy: no launch


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_compute_fails_loudly_without_gpu(L):
    from lorastencil_amd import _lib

    a = L.reference_input("star2d1r", (32, 64))
    with pytest.raises(L.LoraError) as e:
        L.run_host("star2d1r", a, times=1)
    assert e.value.status == _lib.LORA_ENODEVICE


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under lorastencil_amd/ may import, load or link it."""
    pkg = os.path.join(ROOT, "lorastencil_amd")
    pat = re.compile(r"(from\s+oracle|import\s+oracle|liblorastencil_oracle|oracle/|lorastencil_oracle\.h|-loracle)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{os.path.join(dirpath, f)} reaches into oracle/"


def test_diagnostics_are_not_in_the_shipped_library(L, monkeypatch):
    """`ablate` (loads / stores removed: wrong results by construction) exists only in -DLORA_DIAGNOSTICS builds, and no
    environment variable can swap the engine library."""
    from lorastencil_amd import _lib

    p = L.Plan("star2d1r", (64, 128))
    with pytest.raises(L.LoraError) as e:
        p.set_option("ablate", 1)
    assert e.value.status == _lib.LORA_EINVAL
    with pytest.raises(L.LoraError):
        p.get_option("ablate")
    src = open(os.path.join(ROOT, "lorastencil_amd", "_lib.py")).read()
    assert "os.environ" not in src
    assert _lib.LIB_PATH == os.path.join(ROOT, "lorastencil_amd", "lib", "liblorastencil_hip.so")


def test_kernel_signature_names_every_selecting_option(L):
    p = L.Plan("star2d1r", (16384, 16384))
    assert p.kernel_name == "stencil2d_wg_kernel"  # the default 2D kernel: workgroup rows, six applications per launch
    assert p.get_option("steps_per_launch") == 6
    assert p.kernel_signature.startswith(p.kernel_name + "[") and "eval=7" in p.kernel_signature and "k=6" in p.kernel_signature
    s6 = p.kernel_signature
    p.set_option("wg_edge_pct", 20)
    assert p.kernel_signature != s6
    # plain 49-tap tables (no low-rank form) run six applications per launch too (eval = 2: the taps one by one); the
    # Dirichlet option runs four
    q = L.Plan("star2d1r", (256, 512)).set_weights(np.random.default_rng(3).random(49))
    assert q.get_option("steps_per_launch") == 6 and q.kernel_name == "stencil2d_wg_kernel" and "eval=2" in q.kernel_signature
    assert L.Plan("star2d1r", (256, 512)).set_boundary("dirichlet").get_option("steps_per_launch") == 4
    p = L.Plan("star2d1r", (16384, 16384)).set_option("steps_per_launch", 4)
    assert p.kernel_name == "stencil2d_stream_kernel"  # row-streaming, four applications
    assert p.kernel_signature.startswith(p.kernel_name + "[") and "eval=7" in p.kernel_signature
    assert "k=4" in p.kernel_signature and "depth=" in p.kernel_signature and "rows=" in p.kernel_signature
    s0 = p.kernel_signature
    p.set_option("lowrank_valu", 0)
    assert p.kernel_signature != s0  # another instantiation: measured traffic filed under s0 no longer applies
    p.set_option("stream", 0)
    assert p.kernel_name == "stencil2d_fused2_kernel" and "rows=10" in p.kernel_signature
    assert L.Plan("1d1r", (4096,)).kernel_signature == "stencil1d_fusedk_kernel[k=8]"
    with pytest.raises(L.LoraError):
        L.Plan("1d1r", (2**31 - 8,))  # padded extent n + 8 would overflow the kernels' 32-bit indices


@pytest.mark.parametrize("shape,dims,world", [
    ("star3d1r", (208, 384, 384), 4),   # end ranks resolved (2, 4), middle ranks (4, 8)
    ("star3d1r", (160, 384, 384), 3),
    ("star3d1r", (496, 256, 256), 4),
    ("box3d1r", (300, 384, 384), 4),
    ("star3d1r", (512, 512, 512), 8),
    ("star2d1r", (16384, 16384), 8),
    ("box2d3r", (8192, 8191), 4),       # odd innermost extent
    ("1d1r", (1 << 20,), 8),
])
def test_slab_schedule_is_the_same_on_every_rank(L, shape, dims, world):
    """The launch depth, ghost need and refresh interval of a slab run come from the GLOBAL grid and are forced on every
    rank's local plan (lorastencil_amd/slab.py: slab_schedule): replay the selection for each rank of decompositions
    whose local grids straddle the plans' size crossovers and require identical answers.  Host logic only."""
    from lorastencil_amd import ops, slab

    sid = ops.shape_id(shape)
    make = lambda lay: slab.HipStepper(lay)
    seen = set()
    for rank in range(world):
        sched = slab.slab_schedule(sid, dims, world, rank, False, make, True, None)
        assert sched is not None
        layout, stepper, apps, need, every = sched
        assert stepper.apps_per_launch == apps, (rank, stepper.apps_per_launch, apps)
        assert layout.ghost == need * every
        seen.add((apps, need, every, layout.ghost))
    assert len(seen) == 1, seen


@pytest.mark.parametrize("team", [0, 1, 2])
def test_spans_cover_every_tile_plane_exactly_once(L, team):
    """csrc/spans.h: a launch of the register-resident 3D kernels cut into spans / team spans.  The decode the kernels run
    is replayed on the host (lora_debug_span_cover): whatever the tile grid, depth, start length and number of resident
    workgroups, every (tile, plane) pair is swept by exactly one segment, no workgroup decodes a pair outside the region,
    and the busiest workgroup stays within two starts of an even share."""
    import ctypes

    from lorastencil_amd import _lib

    lib = _lib.lib()
    cases = [(7, 14, 768, 11, 256), (5, 22, 512, 9, 256), (5, 22, 64, 9, 256), (1, 1, 1, 9, 256), (2, 2, 5, 5, 256), (3, 3, 7, 11, 8),
             (9, 19, 1024, 11, 256), (7, 32, 96, 9, 256), (1, 40, 300, 11, 304), (40, 1, 33, 5, 64), (3, 5, 1000, 9, 7), (4, 4, 2, 11, 256),
             (13, 3, 17, 9, 100)]
    for tx, ty, depth, S, slots in cases:
        cover = (ctypes.c_int * (tx * ty * depth))()
        wgs, busiest = ctypes.c_int(), ctypes.c_int()
        rc = lib.lora_debug_span_cover(tx, ty, depth, S, slots, team, cover, ctypes.byref(wgs), ctypes.byref(busiest))
        if team and tx > slots:
            assert rc == L.LORA_EUNSUPPORTED if hasattr(L, "LORA_EUNSUPPORTED") else rc != 0
            continue
        assert rc == 0, (tx, ty, depth, S, slots)
        counts = np.frombuffer(cover, dtype=np.int32)
        assert counts.min() == 1 and counts.max() == 1, (tx, ty, depth, S, slots, team, int(counts.min()), int(counts.max()))
        assert 1 <= wgs.value <= slots
        if team == 2:  # (pieces per column differ: the rim columns' are shorter; the bound is on a column between them)
            assert busiest.value <= 1.4 * ty * (depth + S) / max(1, (slots - 2) // max(tx, 1)) + 2 * S + 2 or tx < 3
            continue
        groups = wgs.value // tx if team else wgs.value  # independent pieces of the line
        lines = ty if team else tx * ty
        even = lines * (depth + S) / groups
        assert busiest.value <= 1.25 * even + 2 * S + 2, (tx, ty, depth, S, slots, team, busiest.value, even)
