"""Parity tests proper (GPU): the HIP path, called through the C ABI, against the golden fixtures (generated
from the reference's own test_cpu), against the oracle on seeded inputs, and -- at BASELINE.json's full sizes --
through size-independent properties.

Tolerances: the grids hold small integers, so every partial sum is an exactly representable integer while
|values| < 2^53: results are then BIT-EXACT whatever the summation order or FMA contraction (steps <= 4 here).
Longer runs and random real weights are held to the north star's 1e-10 relative bound (measured ~1e-15)."""
import os
import subprocess

import numpy as np
import pytest
from conftest import ALL_SHAPES, ROOT, load_golden

pytestmark = pytest.mark.gpu

REL_TOL = 1e-10  # BASELINE.json north_star: "output within 1e-10 relative of reference"


@pytest.fixture(scope="module")
def L(engine_built):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import lorastencil_amd as L

    assert os.path.exists(L._lib.LIB_PATH)
    assert L.device_count() >= 1
    return L


@pytest.fixture(scope="module")
def O():
    from oracle import oracle as O

    return O


def rel_err(got, exp):
    """Point-wise relative error with an absolute floor: |got - exp| / max(|exp|, 1e-2 x the largest |exp|).  (A plain
    max-norm ratio lets every point hide behind the largest one; the floor keeps points of signed data that cancel to
    nearly nothing -- whose rounding error is set by the size of their addends, not of their sum -- from reading as
    large relative errors.)"""
    exp = np.asarray(exp, dtype=np.float64)
    scale = np.abs(exp).max()
    if not scale > 0:
        return float(np.abs(got - exp).max())
    return float((np.abs(got - exp) / np.maximum(np.abs(exp), 1e-2 * scale)).max())


def plan_run(L, shape, a, times, weights=None, options=None, params=None):
    """Device-resident path: buffers are torch tensors, sweeps go through lora_plan_run."""
    import torch

    h = L.ops.halo(shape)
    dims = tuple(a.shape[i] - 2 * h[i] for i in range(a.ndim))
    plan = L.Plan(shape, dims, params)
    if weights is not None:
        plan.set_weights(weights)
    for k, v in (options or {}).items():
        plan.set_option(k, v)
    b0 = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, times)
    torch.cuda.synchronize()
    return (b0, b1)[times % 2].cpu().numpy()


def plan_run_variant(L, shape, a, times, variant, weights=None):
    import torch

    h = L.ops.halo(shape)
    dims = tuple(a.shape[i] - 2 * h[i] for i in range(a.ndim))
    plan = L.Plan(shape, dims)
    if weights is not None:
        plan.set_weights(weights)
    plan.set_variant(variant)
    assert plan.get_option("variant") == variant
    assert "mfma" in plan.kernel_name or variant != L.VARIANT_MFMA
    b0 = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, times)
    torch.cuda.synchronize()
    return (b0, b1)[times % 2].cpu().numpy()


# ---------------------------------------------------------------------------------------------------------
# golden fixtures (reference test_cpu) through both C-ABI groups
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ALL_SHAPES)
def test_host_operator_matches_golden_bit_exact(L, shape):
    g = load_golden(shape)
    for t in (1, 2, 3, 4):
        out, info = L.run_host(shape, g["input"], params=g["params"], times=t)
        exp = g[f"out_t{t}"]
        if out.ndim == 1:
            assert np.array_equal(out[:-1], exp[:-1]) and out[-1] == 0.0  # 1d/gpu_1r.cu:134
        else:
            assert np.array_equal(out, exp), f"{shape} t={t}"
        assert info.sweep_seconds > 0 and info.gstencils > 0


@pytest.mark.parametrize("shape", ALL_SHAPES)
def test_plan_path_matches_golden_bit_exact(L, shape):
    g = load_golden(shape)
    for t in (0, 1, 4):
        exp = g[f"out_t{t}"] if t else g["input"]
        assert np.array_equal(plan_run(L, shape, g["input"], t), exp), f"{shape} t={t}"


def test_reference_named_operators(L, capfd):
    """gpu_star_2d1r(in, out, params, times, m, n) etc.: same call shape as the reference, same stdout lines."""
    g = load_golden("star2d1r")
    out = np.zeros_like(g["input"])
    L.gpu_star_2d1r(g["input"], out, g["params"], 2, 64, 128)
    assert np.array_equal(out, g["out_t2"])
    lines = capfd.readouterr().out.splitlines()
    assert lines[0] == "LoRAStencil(2D star_2d1r): " and lines[1].startswith("Time = ") and lines[1].endswith("[ms]")
    assert lines[2].startswith("GStencil/s = ")
    g = load_golden("box3d1r")
    out = np.zeros_like(g["input"])
    L.gpu_box_3d1r(g["input"], out, g["params"], 1, 8, 16, 128)
    assert np.array_equal(out, g["out_t1"])
    g = load_golden("1d2r")
    out = np.zeros_like(g["input"])
    L.gpu_1d2r(g["input"], out, g["params"], 3, 2048)
    assert np.array_equal(out[:-1], g["out_t3"][:-1])
    g = load_golden("box2d1r")  # box2d1r is served by gpu_box_2d3r (2d/main.cu:276-279)
    out = np.zeros_like(g["input"])
    L.gpu_box_2d3r(g["input"], out, g["params"], 1, 32, 64)
    assert np.array_equal(out, g["out_t1"])


# ---------------------------------------------------------------------------------------------------------
# oracle on seeded inputs: ragged sizes, kernel options, long runs, real weights
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,dims", [
    ("star2d1r", (32, 64)), ("star2d1r", (40, 130)), ("star2d1r", (1, 2)), ("star2d1r", (33, 254)),
    ("box2d3r", (96, 256)), ("box2d3r", (7, 6)), ("star2d3r", (50, 386)), ("box2d1r", (64, 64)),
    ("star3d1r", (3, 5, 6)), ("star3d1r", (17, 16, 128)), ("star3d1r", (33, 19, 258)), ("box3d1r", (16, 33, 130)),
    ("box3d1r", (1, 1, 2)), ("1d1r", (1024,)), ("1d1r", (1027,)), ("1d2r", (2,)), ("1d2r", (70001,)),
])
def test_ragged_sizes_match_oracle(L, O, shape, dims):
    a = O.reference_input(shape, dims)
    for t in (1, 3):
        got = plan_run(L, shape, a, t)
        exp = O.run(shape, a, t)
        if a.ndim == 1:
            exp[-1] = got[-1]
        assert np.array_equal(got, exp), f"{shape} {dims} t={t}"


@pytest.mark.parametrize("shape,dims", [("star2d1r", (33, 65)), ("box2d3r", (20, 131)), ("star2d3r", (1, 1)),
                                        ("star3d1r", (7, 9, 33)), ("box3d1r", (3, 2, 1))])
def test_odd_innermost_extents_use_the_generic_kernels(L, O, shape, dims):
    """Rows of an odd innermost extent are only 8-byte aligned: the tiled kernels do not apply.  3D: the generic
    one-thread-per-point kernels.  2D: fused launches still go through the row-streaming kernel (dword-aligned 16-byte
    pieces, last half pair cut by the store descriptor), only single sweeps through the generic kernel."""
    a = O.reference_input(shape, dims)
    plan = L.Plan(shape, dims)
    if len(dims) == 3:
        assert "generic" in plan.kernel_name
    else:
        # (odd extents keep the direct tap order -- the box then evaluates its 49 taps one by one, six applications per
        # launch all the same)
        assert plan.kernel_name == "stencil2d_wg_kernel" and plan.get_option("steps_per_launch") == 6
        p4 = L.Plan(shape, dims).set_option("steps_per_launch", 4)
        assert p4.kernel_name == "stencil2d_stream_kernel" and p4.get_option("steps_per_launch") == 4
        assert "generic" in L.Plan(shape, dims).set_option("stream", 0).kernel_name
        assert "generic" in L.Plan(shape, dims).set_option("steps_per_launch", 1).kernel_name
    for t in (1, 4, 5, 6):
        assert np.array_equal(plan_run(L, shape, a, t), O.run(shape, a, t)), f"{shape} {dims} t={t}"
        assert np.array_equal(plan_run(L, shape, a, t, options={"stream": 0} if len(dims) == 2 else None), O.run(shape, a, t))
    out, _ = L.run_host(shape, a, times=2)
    assert np.array_equal(out, O.run(shape, a, 2))


@pytest.mark.parametrize("shape", ["star2d1r", "box2d3r", "star2d3r"])
@pytest.mark.parametrize("dims", [(33, 65), (20, 131), (1, 1), (7, 3), (64, 127), (300, 1001), (129, 233)])
def test_odd_innermost_extents_through_the_row_streaming_kernel(L, O, shape, dims):
    a = O.reference_input(shape, dims)
    for t in (2, 4, 5, 6, 9, 12, 14):
        for opts in ({}, {"steps_per_launch": 4}, {"steps_per_launch": 2}):
            got, exp = plan_run(L, shape, a, t, options=opts), O.run(shape, a, t)
            if np.abs(exp).max() < 2.0 ** 50:
                assert np.array_equal(got, exp), (shape, dims, t, opts)   # whole padded buffer, halo included
            else:
                assert rel_err(got, exp) < 1e-13, (shape, dims, t, opts)


@pytest.mark.parametrize("rpt", [4, 8, 16])
@pytest.mark.parametrize("panel", [1, 3, 8, 64])
def test_2d_kernel_options_do_not_change_results(L, O, rpt, panel):
    shape, dims = "star2d1r", (160, 640)
    a = O.reference_input(shape, dims)
    exp = O.run(shape, a, 2)
    got = plan_run(L, shape, a, 2, options={"rows_per_thread": rpt, "panel_width": panel})
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("shape", ["star2d1r", "box2d3r", "box2d1r", "star2d3r"])
def test_mfma_lowrank_variant_matches_golden_and_oracle(L, O, shape):
    """The low-rank (U X) V formulation on v_mfma_f64_16x16x4 (kernels_2d_mfma.hip) against the same fixtures."""
    g = load_golden(shape)
    dims = tuple(int(d) for d in g["dims"])
    for t in (1, 2, 4):
        got = plan_run_variant(L, shape, g["input"], t, L.VARIANT_MFMA)
        assert np.array_equal(got, g[f"out_t{t}"]), f"{shape} t={t}"
    for dims in ((40, 130), (96, 384), (1, 2)):  # ragged tiles
        a = O.reference_input(shape, dims)
        assert np.array_equal(plan_run_variant(L, shape, a, 3, L.VARIANT_MFMA), O.run(shape, a, 3)), dims


def test_mfma_variant_real_weights_and_scaling(L, O):
    rng = np.random.default_rng(9)
    a = rng.standard_normal(O.padded_shape("star2d1r", (64, 256)))
    # normalised star2d1r taps (a scaled copy of the reference table keeps its rank-1 + correction form)
    w = O.effective_weights("star2d1r") / 100.0
    got = plan_run_variant(L, "star2d1r", a, 5, L.VARIANT_MFMA, weights=w)
    assert rel_err(got, O.run("star2d1r", a, 5, weights=w)) < 1e-13
    # a symmetric box table with a non-vanishing 4th pyramid term: the residual goes to the vector pipe
    p = O.default_params("box2d3r").copy()
    p[24] = 10.0
    got = plan_run_variant(L, "box2d3r", a, 2, L.VARIANT_MFMA, weights=p / p.sum())
    assert rel_err(got, O.run("box2d3r", a, 2, weights=p / p.sum())) < 1e-13
    # asymmetric rank-2 taps (plus one isolated tap): outside the pyramid scheme, taken through the SVD factoriser
    w2 = np.outer(rng.standard_normal(7), rng.standard_normal(7)) + np.outer(rng.standard_normal(7), rng.standard_normal(7))
    w2[0, 6] += 0.75
    w2 = (w2 / np.abs(w2).sum()).ravel()
    got = plan_run_variant(L, "box2d3r", a, 3, L.VARIANT_MFMA, weights=w2)
    assert rel_err(got, O.run("box2d3r", a, 3, weights=w2)) < 1e-12
    # taps with no low-rank form are refused, not mis-computed
    plan = L.Plan("box2d3r", (64, 256))
    plan.set_weights(rng.standard_normal(49))
    with pytest.raises(L.LoraError):
        plan.set_variant(L.VARIANT_MFMA)


# ---------------------------------------------------------------------------------------------------------
# temporal fusion: K applications per launch must equal K launches.  Four 2D kernels: the workgroup-row kernel with six
# applications per launch (kernels_2d_wg.hip; the default, its tails run four / two through the next one), the
# row-streaming kernel with four (kernels_2d_stream.hip), the same with two, and the tile kernel (kernels_2d_fused.hip, two)
# ---------------------------------------------------------------------------------------------------------
FUSED_2D = {"wg6": {}, "wg6_short": {"wg_rows": 20, "wg_edge_pct": 0}, "stream4": {"steps_per_launch": 4},
            "stream2": {"steps_per_launch": 2}, "tile2": {"stream": 0, "steps_per_launch": 2}}


@pytest.mark.parametrize("kernel", list(FUSED_2D))
@pytest.mark.parametrize("shape,dims", [("star2d1r", (64, 128)), ("star2d1r", (26, 122)), ("star2d1r", (52, 244)),
                                        ("star2d1r", (53, 246)), ("star2d1r", (40, 130)), ("star2d1r", (1, 2)),
                                        ("star2d1r", (300, 700)), ("box2d3r", (64, 128)), ("box2d3r", (90, 250)),
                                        ("star2d3r", (64, 128)), ("star2d3r", (27, 124)), ("star2d1r", (37, 104)),
                                        ("star2d1r", (700, 118)), ("box2d3r", (13, 232)), ("star2d3r", (40, 2100)),
                                        ("star2d1r", (37, 476)), ("star2d3r", (13, 952)), ("box2d3r", (150, 1430))])
def test_fused_two_step_launches_equal_step_by_step(L, O, shape, dims, kernel):
    a = O.reference_input(shape, dims)
    if kernel.startswith("wg6"):
        plan = L.Plan(shape, dims)
        assert plan.kernel_name == "stencil2d_wg_kernel" and plan.get_option("steps_per_launch") == 6
    # 12, 13: three four-sweep launches, the last two through the scratch grid; 18 .. 23: three six-sweep launches and
    # every tail (4, 2, 4 + 1 ...), odd launch counts through the scratch grid
    for t in (4, 5, 6, 7, 8, 9, 10, 11, 12, 13) + ((16, 17, 18, 20, 22, 23) if kernel.startswith("wg6") else ()):
        got = plan_run(L, shape, a, t, options=FUSED_2D[kernel])
        exp = O.run(shape, a, t)
        # whole padded buffer: interior AND the halo state the step-by-step driver leaves behind
        if np.abs(exp).max() < 2.0 ** 50:  # (structured evaluations run partial sums a few bits ahead of the final value)
            assert np.array_equal(got, exp), f"{shape} {dims} t={t}"
        else:
            assert rel_err(got, exp) < 1e-13, f"{shape} {dims} t={t}"


@pytest.mark.parametrize("dims", [(64, 128), (90, 250), (13, 233), (150, 1430), (700, 118)])
def test_general_49_tap_tables_run_six_applications_per_launch(L, O, dims):
    """A 7 x 7 table with no low-rank form (the reference's gpu_box_2d3r honours all 49 params, 2d/gpu.cu:276-350): the
    workgroup-row kernel with the taps evaluated one by one, six applications per launch by default (432 against 333
    GStencils/s with four at 16384^2), its four / two tails and single sweeps -- exact on small integers, to rounding on
    real data."""
    rng = np.random.default_rng(dims[0])
    wi = rng.integers(-2, 3, 49).astype(np.float64) / 64.0  # full rank with probability ~1: no factor form; dyadic: exact sums
    a = np.zeros(O.padded_shape("box2d3r", dims))
    a[...] = rng.integers(-2, 3, a.shape)
    plan = L.Plan("box2d3r", dims).set_weights(wi)
    assert plan.kernel_name == "stencil2d_wg_kernel" and plan.get_option("steps_per_launch") == 6
    for t in (6, 7, 8):  # (numerators below 2^52: 49 taps of |w| <= 2 grow a sweep's values by at most 64)
        exp = O.run("box2d3r", a, t, weights=wi)
        assert np.array_equal(plan_run(L, "box2d3r", a, t, weights=wi), exp), (dims, t)
    wr = rng.standard_normal(49)
    wr /= np.abs(wr).sum()
    b = rng.standard_normal(a.shape)
    for t in (6, 13, 23):
        assert rel_err(plan_run(L, "box2d3r", b, t, weights=wr), O.run("box2d3r", b, t, weights=wr)) < 1e-12, (dims, t)


@pytest.mark.parametrize("shape,dims,times", [("star2d1r", (150, 380), 20), ("box2d3r", (70, 130), 12), ("star3d1r", (20, 24, 64), 10),
                                              ("box3d1r", (12, 16, 64), 6), ("1d1r", (20000,), 27)])
def test_odd_numbers_of_fused_launches_with_and_without_the_scratch_grid(L, O, shape, dims, times):
    """lora_plan_run with an odd number of fused launches: routed through the plan's scratch grid (default) or, with
    option scratch = 0, rescheduled to an even number -- same result, whole padded buffer, as the step-by-step driver."""
    w = O.effective_weights(shape)
    w = w / w.sum()
    a = O.reference_input(shape, dims)
    exp = O.run(shape, a, times, weights=w)
    for scratch in (1, 0):
        got = plan_run(L, shape, a, times, weights=w, options={"scratch": scratch})
        if a.ndim == 1:
            got[-1] = exp[-1]
        assert rel_err(got, exp) < 1e-13, (shape, scratch)
        h = L.ops.halo(shape)
        edge = tuple(slice(0, k) for k in h)
        assert np.array_equal(got[edge], exp[edge])


@pytest.mark.parametrize("shape,dims", [("star2d1r", (300, 700)), ("star2d1r", (1, 2)), ("star2d3r", (2100, 2600)),
                                        ("box2d3r", (90, 250))])
def test_fused_persistent_workgroups(L, O, shape, dims):
    """persistent = 1: 2 workgroups per CU walk the tiles and prefetch the next window into registers."""
    a = O.reference_input(shape, dims)
    for t in (4, 5):
        got = plan_run(L, shape, a, t, options={"stream": 0, "steps_per_launch": 2, "persistent": 1})
        assert np.array_equal(got, O.run(shape, a, t)), f"{shape} {dims} t={t}"


@pytest.mark.parametrize("rows", [6, 8, 10])
def test_fused_tile_heights(L, O, rows):
    for shape, dims in (("star2d1r", (150, 380)), ("box2d3r", (70, 130))):
        a = O.reference_input(shape, dims)
        got = plan_run(L, shape, a, 4, options={"stream": 0, "steps_per_launch": 2, "fused_rows": rows})
        assert np.array_equal(got, O.run(shape, a, 4)), (shape, rows)


def test_fused_lowrank_evaluation_on_the_vector_pipe(L, O):
    """The fused kernel evaluates star2d1r / box tables through their low-rank factors (horizontal pass per term,
    then vertical scatter) when the factors fit; forcing the direct taps must give the same grid."""
    import torch

    for shape, expect_eval in (("star2d1r", 7), ("box2d3r", 6), ("box2d1r", 6), ("star2d3r", 1)):
        dims = (150, 380)
        a = O.reference_input(shape, dims)
        exp = O.run(shape, a, 4)
        plan = L.Plan(shape, dims)
        assert plan.get_option("fused_eval") == expect_eval, shape
        for k in FUSED_2D.values():
            assert np.array_equal(plan_run(L, shape, a, 4, options=k), exp), shape
            assert np.array_equal(plan_run(L, shape, a, 4, options={**k, "lowrank_valu": 0}), exp), shape
            assert np.array_equal(plan_run(L, shape, a, 4, options={**k, "lowrank_valu": 2}), exp), shape  # plain pyramid form
            assert np.array_equal(plan_run(L, shape, a, 4, options={**k, "lowrank_valu": 3}), exp), shape  # symmetric, no gap
            assert np.array_equal(plan_run(L, shape, a, 4, options={**k, "lowrank_valu": 4}), exp), shape  # rank 1 + correction
        off = L.Plan(shape, dims).set_option("lowrank_valu", 0)
        assert off.get_option("fused_eval") == {"star2d1r": 0, "box2d3r": 2, "box2d1r": 2, "star2d3r": 1}[shape]
    # scaled star2d1r taps keep the form; real-valued data within rounding of the oracle
    rng = np.random.default_rng(8)
    a = rng.standard_normal(O.padded_shape("star2d1r", (128, 512)))
    w = O.effective_weights("star2d1r") / 100.0
    plan = L.Plan("star2d1r", (128, 512)).set_weights(w)
    assert plan.get_option("fused_eval") == 7  # nested-profile form: rows (1), 2 (1 2 1), 2 (1 2 4 2 1), (1 4 8 16 8 4 1)
    assert L.Plan("star2d1r", (128, 512)).set_weights(w).set_option("lowrank_valu", 4).get_option("fused_eval") == 3
    for lr in (-1, 4):
        got = plan_run(L, "star2d1r", a, 8, weights=w, options={"lowrank_valu": lr})
        assert rel_err(got, O.run("star2d1r", a, 8, weights=w)) < 1e-13
    assert rel_err(plan_run(L, "star2d1r", a, 8, weights=w), O.run("star2d1r", a, 8, weights=w)) < 1e-13
    # a rank-4 box table (centre 10): the residual tap rules the pyramid form out, direct taps are used
    p4 = O.default_params("box2d3r").copy()
    p4[24] = 10.0
    plan = L.Plan("box2d3r", (64, 128)).set_weights(p4 / p4.sum())
    assert plan.get_option("fused_eval") == 2
    # general taps: direct
    assert L.Plan("box2d3r", (64, 128)).set_weights(rng.standard_normal(49)).get_option("fused_eval") == 2


def test_fused_step2_direct_call_and_regions(L, O):
    import torch

    shape, dims = "star2d1r", (200, 380)
    a = O.reference_input(shape, dims)
    exp = O.run(shape, a, 2)  # buffer 0 after two sweeps: interior + the input halo
    assert L.Plan(shape, dims).kernel_name == "stencil2d_wg_kernel"
    assert L.Plan(shape, dims).get_option("steps_per_launch") == 6
    assert L.Plan(shape, dims).set_option("steps_per_launch", 4).kernel_name == "stencil2d_stream_kernel"
    pd = L.Plan(shape, dims).set_boundary("dirichlet")  # four per launch: the workgroup-row kernel with halo values at every level
    assert pd.get_option("steps_per_launch") == 4 and pd.kernel_name == "stencil2d_wg_kernel"
    assert L.Plan(shape, dims).set_boundary("dirichlet").set_option("wg", 0).get_option("steps_per_launch") == 2
    assert L.Plan(shape, dims).set_option("stream", 0).get_option("steps_per_launch") == 2
    _check_step2_and_regions(L, O, shape, dims, a, exp, {"stream": 0}, "stencil2d_fused2_kernel")
    _check_step2_and_regions(L, O, shape, dims, a, exp, {}, "stencil2d_stream_kernel")
    # six / four applications in one call: lora_plan_stepk, whole grid and regions in any order
    for k_apps in (6, 4):
        expk = O.run(shape, a, k_apps)
        plan = L.Plan(shape, dims).set_option("steps_per_launch", k_apps)
        src = torch.from_numpy(a).cuda()
        dst = torch.from_numpy(a).cuda()
        dst[4:-4, 4:-4] = -1.0
        plan.stepk(src, dst)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), expk), k_apps
        dst[4:-4, 4:-4] = -1.0
        for b, e in ((100, 200), (0, 26), (26, 100)):
            plan.stepk_region(src, dst, b, e)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), expk), k_apps
    with pytest.raises(L.LoraError):
        L.Plan("star2d1r", (64, 64)).set_option("steps_per_launch", 8)  # 2D fuses two, four or six applications
    with pytest.raises(L.LoraError):
        L.Plan("star3d1r", (8, 8, 64)).set_option("steps_per_launch", 8)  # 3D kernels fuse two, three or four (fp64)
    with pytest.raises(L.LoraError):
        L.Plan("1d1r", (64,)).set_option("steps_per_launch", 3)


def _check_step2_and_regions(L, O, shape, dims, a, exp, opts, name):
    import torch

    plan = L.Plan(shape, dims)
    for k, v in opts.items():
        plan.set_option(k, v)
    assert plan.set_option("steps_per_launch", 2).kernel_name == name
    src = torch.from_numpy(a).cuda()
    dst = torch.from_numpy(a).cuda()  # same halo as the source, interior to be overwritten
    dst[4:-4, 4:-4] = -1.0
    plan.step2(src, dst)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), exp)
    dst[4:-4, 4:-4] = -1.0
    for b, e in ((100, 200), (0, 26), (26, 100)):  # regions in any order
        plan.step2_region(src, dst, b, e)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), exp)


@pytest.mark.parametrize("shape", ["1d1r", "1d2r"])
@pytest.mark.parametrize("n", [1048576, 5000, 4097, 30, 2048, 2050])
def test_1d_fused_k_step_launches_equal_step_by_step(L, O, shape, n):
    """stencil1d_fusedk_kernel: 8 (4, 2) applications per launch with the intermediate levels in LDS; halo cells of
    odd intermediate levels are 0 and of even ones the input halo, as the step-by-step driver leaves them."""
    import torch

    dims = (n,)
    a = O.reference_input(shape, dims)
    assert L.Plan(shape, dims).kernel_name == "stencil1d_fusedk_kernel"
    assert L.Plan(shape, dims).set_option("steps_per_launch", 1).kernel_name == "stencil1d_kernel"
    for k in (32, 16, 8, 4, 2, 0):  # 0: the run-length dependent depth of lora_plan_run (8 / 16 / 32)
        for t in ((2 * k, 2 * k + 1, 3 * k, 4 * k + 3) if k else (17, 40, 67, 100, 131)):
            exp = O.run(shape, a, t)[:-1]  # the host operator's copy-back omits the last element
            got = plan_run(L, shape, a, t, options={"steps_per_launch": k})[:-1]
            if np.abs(exp).max() < 2.0 ** 50:
                assert np.array_equal(got, exp), (shape, n, k, t)
            else:
                assert rel_err(got, exp) < 1e-13, (shape, n, k, t)
    # real-valued data, normalised taps, long run through the hipGraph path; Dirichlet option
    rng = np.random.default_rng(n)
    b = rng.standard_normal(n + 8)
    w = O.effective_weights(shape)
    w = w / w.sum()
    assert rel_err(plan_run(L, shape, b, 50, weights=w)[:-1], O.run(shape, b, 50, weights=w)[:-1]) < 1e-13
    plan = L.Plan(shape, dims).set_weights(w).set_boundary("dirichlet")
    b0 = torch.from_numpy(b).cuda()
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, 35)
    torch.cuda.synchronize()
    assert rel_err((b0, b1)[35 % 2].cpu().numpy(), O.run_bc(shape, b, 35, "dirichlet", weights=w)) < 1e-13
    # one direct K-step launch on point ranges == K single sweeps (the source buffer carries the level-0 halo)
    plan = L.Plan(shape, dims).set_weights(w)
    src = torch.from_numpy(b).cuda()
    dst = src.clone()
    cuts = sorted({0, (n // 3) & ~1, (2 * n // 3) & ~1, n})
    for lo, hi in list(zip(cuts[:-1], cuts[1:]))[::-1]:
        plan.stepk_region(src, dst, lo, hi)
    torch.cuda.synchronize()
    exp = O.run(shape, b, 8, weights=w)  # buffer 0 after 8 sweeps: the input's halo, new interior
    assert rel_err(dst.cpu().numpy()[:-1], exp[:-1]) < 1e-14
    assert np.array_equal(dst.cpu().numpy()[:4], b[:4]) and np.array_equal(dst.cpu().numpy()[-4:], b[-4:])


# fused 3D fp64 launches: the plane-streaming kernel (three / two applications, its workgroup shapes and the one-barrier
# pipeline) and the round-1 tile kernel
FUSED_3D = {
    "default": {},  # grids this small: the tile kernel (the plane-streaming kernel takes over from ~2.4e7 points)
    "always": {"stream3": 1},
    "stream3": {"steps_per_launch": 3},
    "stream3_w4": {"steps_per_launch": 3, "stream3_waves": 4},
    "stream3_pipe": {"steps_per_launch": 3, "stream3_pipe": 1},
    "stream2": {"stream3": 1, "steps_per_launch": 2},
    "stream2_w4": {"stream3": 1, "steps_per_launch": 2, "stream3_waves": 4},
    "stream2_w4pipe": {"stream3": 1, "steps_per_launch": 2, "stream3_waves": 4, "stream3_pipe": 1},
    "tile2": {"stream3": 0, "steps_per_launch": 2},
    "async3": {"steps_per_launch": 3, "stream3_async": 1},
    "async3_w4": {"steps_per_launch": 3, "stream3_async": 1, "stream3_waves": 4},
    "async2": {"stream3": 1, "steps_per_launch": 2, "stream3_async": 1},
}


@pytest.mark.parametrize("shape", ["star3d1r", "box3d1r"])
@pytest.mark.parametrize("dims", [(40, 60, 128), (9, 31, 62), (37, 29, 190), (3, 5, 2), (70, 64, 64)])
@pytest.mark.parametrize("cfg", list(FUSED_3D))
def test_3d_fused_launches_equal_step_by_step(L, O, shape, dims, cfg):
    """kernels_3d_planes.hip (three or two applications per launch, inner time levels in LDS) and kernels_3d_fused.hip
    (two).  Same taps in the same order at every level, so the whole padded buffer (interior and the halo state the
    step-by-step driver leaves behind) equals the oracle bit for bit while values are exact integers, and to rounding
    afterwards.  Three-application launches run on the reference's alternating buffer state: any step count, odd
    numbers of launches included."""
    a = O.reference_input(shape, dims)
    opts = FUSED_3D[cfg]
    plan = L.Plan(shape, dims)
    for k, v in opts.items():
        plan.set_option(k, v)
    if cfg == "default":
        assert plan.get_option("steps_per_launch") == 2
    elif cfg == "always":
        assert plan.get_option("steps_per_launch") == 3  # the star, and the box through its separable form
    else:
        assert plan.get_option("steps_per_launch") == opts["steps_per_launch"]
    assert plan.kernel_name == ("stencil3d_fused2_kernel" if cfg in ("tile2", "default") else "stencil3d_planes_kernel")
    for t in (3, 4, 5, 6, 7, 9):
        exp = O.run(shape, a, t)
        for zc in (0, 1, 3, 8):
            got = plan_run(L, shape, a, t, options=dict(opts, fused_z_chunk=zc))
            if np.abs(exp).max() < 2.0 ** 50:
                assert np.array_equal(got, exp), f"{shape} {dims} {cfg} t={t} zc={zc}"
            else:
                assert rel_err(got, exp) < 1e-13, f"{shape} {dims} {cfg} t={t} zc={zc}"


@pytest.mark.parametrize("shape", ["star3d1r", "box3d1r"])
@pytest.mark.parametrize("dims", [(40, 60, 128), (9, 31, 62), (37, 29, 190), (3, 5, 2), (70, 64, 64), (12, 24, 121), (33, 47, 255),
                                  (30, 100, 260), (1, 1, 1)])
def test_3d_register_resident_kernel_equals_step_by_step(L, O, shape, dims):
    """kernels_3d_lanes.hip: FOUR applications per launch with the time levels in registers (x-neighbours by DPP, edge rows
    through LDS, z streamed), its two-application form for the tails, even and ODD innermost extents, one and many tiles /
    z-chunks.  Taps arrive in the oracle's order at every level: the whole padded buffer -- interior and the halo state the
    step-by-step driver leaves behind -- equals the oracle bit for bit while values are exact integers."""
    a = O.reference_input(shape, dims)
    plan = L.Plan(shape, dims).set_option("steps_per_launch", 4)
    assert plan.get_option("steps_per_launch") == 4 and plan.kernel_name == "stencil3d_lanes_kernel"
    assert L.Plan(shape, dims).set_option("lanes3", 1).kernel_name == "stencil3d_lanes_kernel"
    assert L.Plan(shape, dims).set_boundary("dirichlet").set_option("lanes3", 1).kernel_name != "stencil3d_lanes_kernel"
    for t in (4, 5, 6, 7, 8, 9, 12, 13):  # 12, 13: three launches of four, the last two through the scratch grid
        exp = O.run(shape, a, t)
        # every way a launch is cut along z: by the plan's own rule, fixed chunks, spans (spans.h: equal pieces of the line
        # of all (tile, plane) pairs, workgroups that cross from one tile into the next), the model's chunks, team spans
        # (the line over tile rows, a piece per team of a row's workgroups)
        for cut in ({}, {"fused_z_chunk": 3}, {"fused_z_chunk": 9}, {"fused_z_chunk": 40}, {"spans3": 1}, {"spans3": 0}, {"spans3": 2}):
            got = plan_run(L, shape, a, t, options=dict({"steps_per_launch": 4}, **cut))
            if np.abs(exp).max() < 2.0 ** 50:
                assert np.array_equal(got, exp), f"{shape} {dims} t={t} {cut}"
            else:
                assert rel_err(got, exp) < 1e-13, f"{shape} {dims} t={t} {cut}"


@pytest.mark.parametrize("shape,dims", [("star3d1r", (200, 600, 760)), ("box3d1r", (200, 600, 760)), ("star3d1r", (512, 512, 512)),
                                        ("box3d1r", (96, 250, 1000))])
def test_3d_register_resident_kernel_launch_shapes_at_size(L, O, shape, dims):
    """The ways a big launch of the fp64 register-resident kernel is laid out -- chunks DEALT to the resident workgroups when
    there are more chunks than CUs (200 x 600 x 760: 175 tiles), one chunk more for the tiles of the first tile column and
    row when a single round has CUs to spare (512^3) -- against single sweeps: runs of four, six and nine sweeps, whole
    padded buffer, bit for bit on small integers (every cut of the launch must give the same bits too)."""
    import torch

    ps = L.padded_shape(shape, dims)
    gen = torch.Generator(device="cuda").manual_seed(99)
    a = torch.randint(0, 4, ps, generator=gen, device="cuda").to(torch.float64)
    w = O.effective_weights(shape)  # integer taps on small integers: exact while 36^t x 3 < 2^53

    def run(t, opts):
        plan = L.Plan(shape, dims).set_weights(w)
        for k, v in opts.items():
            plan.set_option(k, v)
        b0, b1 = a.clone(), torch.zeros_like(a)
        plan.run(b0, b1, t)
        torch.cuda.synchronize()
        return (b0, b1)[t % 2], plan.kernel_name

    for t in (4, 6, 9):  # (6 = four + a two-application tail: the plane-streaming kernel on regions of 128 planes and more)
        ref, _ = run(t, {"steps_per_launch": 1})
        assert float(ref.abs().max()) < 2.0 ** 53
        for opts in ({"steps_per_launch": 4}, {"steps_per_launch": 4, "spans3": 1}, {"steps_per_launch": 4, "fused_z_chunk": 24}):
            got, kn = run(t, opts)
            assert kn == "stencil3d_lanes_kernel"
            assert torch.equal(got, ref), (shape, dims, t, opts)
            del got
        del ref


@pytest.mark.parametrize("shape,dtype", [("box3d1r", "f64"), ("star3d1r", "f64"), ("box3d1r", "bf16")])
def test_3d_register_resident_kernel_is_deterministic_under_repetition(L, O, shape, dtype):
    """Many repetitions of one fused run give the same bits every time, under every cut of the launch.  (Round 4: the fp64
    separable path's first steps run one level alone and had no barrier between a wave's rewrite of its edge rows and its
    neighbours' reads of the step before -- errors of 1e-3 in one run of ~200, never twice the same; this test is the watch
    on that class of bug: workgroups that run several segments are the ones that start most often.)"""
    import torch

    dims = (150, 110, 376)
    rng = np.random.default_rng(5)
    w = O.effective_weights(shape)
    w = w / w.sum()
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
    a = torch.from_numpy(rng.standard_normal(O.padded_shape(shape, dims))).to(tdt).cuda()
    it = torch.int16 if dtype == "bf16" else torch.int64
    for cut in (2, 1, -1):
        plan = L.Plan(shape, dims, dtype=dtype).set_weights(w).set_option("steps_per_launch", 4).set_option("spans3", cut)
        first = None
        for rep in range(40):
            b0, b1 = a.clone(), torch.zeros_like(a)
            plan.run(b0, b1, 8)
            torch.cuda.synchronize()
            if first is None:
                first = b0.clone()
                single = L.Plan(shape, dims, dtype=dtype).set_weights(w).set_option("steps_per_launch", 1)
                c0, c1 = a.clone(), torch.zeros_like(a)
                single.run(c0, c1, 8)
                torch.cuda.synchronize()
                assert rel_err(b0.double().cpu().numpy(), c0.double().cpu().numpy()) < (1e-13 if dtype == "f64" else 1e-30) or \
                    torch.equal(b0.view(it), c0.view(it)), (shape, dtype, cut)
            else:
                assert torch.equal(b0.view(it), first.view(it)), (shape, dtype, cut, rep)


@pytest.mark.parametrize("shape", ["star3d1r", "box3d1r"])
def test_3d_register_resident_kernel_real_data_and_regions(L, O, shape):
    """Random real data and taps: four applications in one launch == four single sweeps bit for bit (star: same tap order;
    the separable box to 1e-13: x / y / z passes sum in another order than the 27-tap single sweep) and the oracle to
    rounding; plane ranges in any order compose (what the slab drivers launch)."""
    import torch

    rng = np.random.default_rng(23)
    dims = (45, 70, 250)
    w = O.effective_weights(shape)
    if shape == "star3d1r":
        w = w * rng.uniform(0.5, 1.5, 27)  # any seven star weights
    w = w / w.sum()
    a = rng.standard_normal(O.padded_shape(shape, dims))
    plan = L.Plan(shape, dims).set_weights(w).set_option("steps_per_launch", 4)
    assert plan.kernel_name == "stencil3d_lanes_kernel"
    src = torch.from_numpy(a).cuda()
    dst = torch.from_numpy(a).cuda()
    dst[1:-1, 2:-2, 4:-4] = -3.0
    plan.stepk(src, dst)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    single = plan_run(L, shape, a, 4, weights=w, options={"steps_per_launch": 1})
    if shape == "star3d1r":
        assert np.array_equal(got, single)
    else:
        assert rel_err(got, single) < 1e-13
    assert rel_err(got, O.run(shape, a, 4, weights=w)) < 1e-13
    dst[1:-1, 2:-2, 4:-4] = -3.0
    for b, e in ((20, 45), (0, 7), (7, 20)):
        plan.stepk_region(src, dst, b, e)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), got)
    # two ranges in one launch (lora_plan_stepn_region2: the two ends a slab driver sweeps behind its deferred wait)
    plan.set_option("spans3", -1)
    for (b0, e0, b1, e1), rest in (((0, 7, 20, 45), (7, 20)), ((41, 45, 0, 4), (4, 41)), ((0, 0, 30, 45), (0, 30))):
        dst[1:-1, 2:-2, 4:-4] = -3.0
        plan.stepn_region2(4, src, dst, b0, e0, b1, e1)
        plan.stepk_region(src, dst, *rest)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), got), (b0, e0, b1, e1)
    for cut in (1, 2, 0):  # the same regions cut into spans / team spans / the model's chunks
        dst[1:-1, 2:-2, 4:-4] = -3.0
        plan.set_option("spans3", cut)
        for b, e in ((7, 20), (20, 45), (0, 7)):
            plan.stepk_region(src, dst, b, e)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), got), cut
    # non-separable 27 taps are not this kernel's: the plan keeps the tile / plane kernels and their two applications
    p27 = L.Plan("box3d1r", dims).set_weights(rng.random(27)).set_option("steps_per_launch", 4)
    assert p27.kernel_name != "stencil3d_lanes_kernel" and p27.get_option("steps_per_launch") == 2


@pytest.mark.parametrize("shape", ["star3d1r", "box3d1r"])
@pytest.mark.parametrize("boundary", ["reference", "dirichlet"])
def test_3d_three_step_launches_real_weights_both_boundaries(L, O, shape, boundary):
    """Three applications per launch on real data, both boundaries, against single sweeps bit for bit (same tap order)
    and against the oracle to rounding; plus the public stepk entry (an even global step: level 1 sees the zero halo,
    level 2 the source's)."""
    import torch

    rng = np.random.default_rng(17)
    dims = (21, 70, 124)
    a = rng.standard_normal(O.padded_shape(shape, dims))
    a[0] = rng.standard_normal(a[0].shape)  # a non-trivial caller halo
    w = rng.standard_normal(27) if shape == "box3d1r" else O.effective_weights(shape) / 8.0
    w = w / np.abs(w).sum()

    def run(times, opts):
        plan = L.Plan(shape, dims).set_weights(w)
        if boundary != "reference":
            plan.set_boundary(boundary)
        for k, v in opts.items():
            plan.set_option(k, v)
        b0 = torch.from_numpy(a).cuda()
        b1 = torch.zeros_like(b0)
        plan.run(b0, b1, times)
        torch.cuda.synchronize()
        return (b0, b1)[times % 2].cpu().numpy(), plan

    for t in (3, 6, 8, 10):
        single, _ = run(t, {"steps_per_launch": 1})
        for opts in ({"steps_per_launch": 3}, {"steps_per_launch": 3, "stream3_waves": 4, "fused_z_chunk": 5},
                     {"steps_per_launch": 3, "stream3_pipe": 1}):
            got, plan = run(t, opts)
            assert plan.kernel_name == "stencil3d_planes_kernel" and plan.get_option("steps_per_launch") == 3
            assert np.array_equal(got, single), (shape, boundary, t, opts)
        exp = O.run_bc(shape, a, t, boundary, weights=w) if boundary != "reference" else O.run(shape, a, t, weights=w)
        assert rel_err(single, exp) < 1e-13
    if boundary == "reference":
        plan = L.Plan(shape, dims).set_weights(w).set_option("steps_per_launch", 3)
        src = torch.from_numpy(a).cuda()
        dst = torch.from_numpy(a).cuda()
        dst[1:-1, 2:-2, 4:-4] = -1.0
        plan.stepk(src, dst)
        torch.cuda.synchronize()
        whole = dst.cpu().numpy()
        exp3 = O.run(shape, a, 3, weights=w)
        assert rel_err(whole[1:-1, 2:-2, 4:-4], exp3[1:-1, 2:-2, 4:-4]) < 1e-13
        assert np.array_equal(whole[0], a[0]) and np.array_equal(whole[:, :2], a[:, :2])  # halo untouched
        dst[1:-1, 2:-2, 4:-4] = -1.0
        for b, e in ((15, 21), (0, 4), (4, 15)):
            plan.stepk_region(src, dst, b, e)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), whole)  # plane ranges == one launch, bit for bit


@pytest.mark.parametrize("shape,dims", [("star3d1r", (200, 310, 372)), ("box3d1r", (150, 250, 300))])
def test_3d_stream_kernel_at_scale_equals_single_sweeps(L, shape, dims):
    """Several workgroups per CU and several rounds of them: a wave overtaking its neighbour (a missing barrier between
    publishing a level and the next overwrite of it) shows only at this scale -- the first two-application form of the
    kernel passed every small case and failed here.  Every workgroup shape, bit for bit against single sweeps."""
    import torch

    w = L.effective_weights(shape)
    w = w / w.sum()
    torch.manual_seed(3)
    a = torch.randn(L.padded_shape(shape, dims), dtype=torch.float64, device="cuda")

    def run(opts, times):
        plan = L.Plan(shape, dims).set_weights(w)
        for k, v in opts.items():
            plan.set_option(k, v)
        b0 = a.clone()
        b1 = torch.zeros_like(b0)
        plan.run(b0, b1, times)
        torch.cuda.synchronize()
        return (b0, b1)[times % 2]

    ref = run({"steps_per_launch": 1}, 6)
    for k in (3, 2):
        for wv in (8, 4):
            for extra in ({}, {"stream3_pipe": 1}, {"stream3_async": 1}):
                if "stream3_async" in extra and wv not in (8, 4):
                    continue
                # the box's exactly separable taps run as x / y / z passes in this kernel (another summation order than
                # the single sweeps' 27 taps): bit-identical with that form switched off, to rounding with it
                got = run(dict({"stream3": 1, "steps_per_launch": k, "stream3_waves": wv, "separable": 0}, **extra), 6)
                assert torch.equal(got, ref), (shape, k, wv, extra)
                if shape == "box3d1r" and "stream3_async" not in extra:
                    got = run(dict({"stream3": 1, "steps_per_launch": k, "stream3_waves": wv}, **extra), 6)
                    assert float((got - ref).abs().max()) <= 1e-13 * float(ref.abs().max()), (shape, k, wv, extra)


@pytest.mark.parametrize("boundary", ["reference", "dirichlet"])
def test_3d_box_separable_form_against_the_oracle(L, O, boundary):
    """The reference's box taps depend on dx only (3d/gpu_box.cu:151-164): exactly separable, and the plane-streaming kernel
    evaluates them as x / y / z passes (9-10 multiply-adds per point instead of 27).  Integer data: bit for bit against
    the oracle's 27-tap order (every partial sum exact); real data and scaled taps: to rounding; option separable = 0
    restores the 27-tap order bit for bit."""
    import torch

    shape, dims = "box3d1r", (21, 70, 124)
    rng = np.random.default_rng(23)
    w = O.effective_weights(shape)

    def run(a, w, times, opts):
        plan = L.Plan(shape, dims).set_weights(w).set_option("stream3", 1)
        if boundary != "reference":
            plan.set_boundary(boundary)
        for k, v in opts.items():
            plan.set_option(k, v)
        b0 = torch.from_numpy(a).cuda()
        b1 = torch.zeros_like(b0)
        plan.run(b0, b1, times)
        torch.cuda.synchronize()
        return (b0, b1)[times % 2].cpu().numpy(), plan

    oracle = (lambda a, w, t: O.run(shape, a, t, weights=w)) if boundary == "reference" else \
             (lambda a, w, t: O.run_bc(shape, a, t, boundary, weights=w))
    a_int = O.reference_input(shape, dims)
    for k in (2, 3):
        got, plan = run(a_int, w, 4, {"steps_per_launch": k})
        assert "taps=2" in plan.kernel_signature and plan.get_option("steps_per_launch") == k
        assert np.array_equal(got, oracle(a_int, w, 4))  # integers below 2^53: exact whatever the order
        a = rng.standard_normal(O.padded_shape(shape, dims))
        wn = w / w.sum()
        for t in (3, 6, 7):
            exp = oracle(a, wn, t)
            got, _ = run(a, wn, t, {"steps_per_launch": k})
            assert rel_err(got, exp) < 1e-13, (k, t)
            direct, plan = run(a, wn, t, {"steps_per_launch": k, "separable": 0})
            assert "taps=1" in plan.kernel_signature
            single, _ = run(a, wn, t, {"steps_per_launch": 1})
            assert np.array_equal(direct, single), (k, t)
    # taps that are not exactly separable keep the 27-tap form
    plan = L.Plan(shape, dims).set_weights(rng.standard_normal(27)).set_option("stream3", 1)
    assert "taps=1" in plan.kernel_signature


def test_3d_fused_step2_regions_and_real_weights(L, O):
    import torch

    rng = np.random.default_rng(5)
    for shape in ("star3d1r", "box3d1r"):
        dims = (33, 47, 132)
        a = rng.standard_normal(O.padded_shape(shape, dims))
        w = rng.standard_normal(27) if shape == "box3d1r" else O.effective_weights(shape) / 8.0
        if shape == "box3d1r":
            w /= np.abs(w).sum()
        exp = O.run(shape, a, 2, weights=w)  # buffer 0 after two sweeps: interior + the input halo
        plan = L.Plan(shape, dims).set_weights(w).set_option("steps_per_launch", 2).set_option("stream3", 1)
        assert plan.kernel_name == "stencil3d_planes_kernel"
        src = torch.from_numpy(a).cuda()
        dst = torch.from_numpy(a).cuda()
        dst[1:-1, 2:-2, 4:-4] = -1.0
        plan.step2(src, dst)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        assert np.array_equal(got[0], a[0]) and np.array_equal(got[:, :2], a[:, :2])  # halo untouched
        assert rel_err(got, exp) < 1e-14, shape
        whole = got.copy()
        dst[1:-1, 2:-2, 4:-4] = -1.0
        for b, e in ((20, 33), (0, 7), (7, 20)):  # plane ranges in any order
            plan.step2_region(src, dst, b, e)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), whole), shape  # regions == one launch, bit for bit
        # long run, odd tail, against the oracle to rounding
        for t in (12, 13):
            got = plan_run(L, shape, a, t, weights=w, options={"steps_per_launch": 2})
            assert rel_err(got, O.run(shape, a, t, weights=w)) < 1e-13, (shape, t)
        # Dirichlet runs fall back to single sweeps (the fused kernel implements the reference boundary only)
        b0 = torch.from_numpy(a).cuda()
        b1 = torch.zeros_like(b0)
        plan.set_boundary("dirichlet").run(b0, b1, 6)
        torch.cuda.synchronize()
        assert rel_err(b0.cpu().numpy(), O.run_bc(shape, a, 6, "dirichlet", weights=w)) < 1e-13


def test_fused_long_run_real_weights(L, O):
    rng = np.random.default_rng(77)
    shape, dims = "star2d1r", (128, 512)
    a = rng.standard_normal(O.padded_shape(shape, dims))
    w = O.effective_weights(shape) / 100.0
    for t in (20, 21, 22, 23):
        for k in FUSED_2D.values():
            got = plan_run(L, shape, a, t, weights=w, options=k)
            assert rel_err(got, O.run(shape, a, t, weights=w)) < 1e-13


@pytest.mark.parametrize("opts", [{"stream_rows": 8}, {"stream_rows": 50, "stream_depth": 2, "stream_sync": 0},
                                  {"stream_sync": 2}, {"stream_depth": 3, "stream_rows": 200},
                                  {"steps_per_launch": 2, "stream_depth": 6, "stream_rows": 29},
                                  {"steps_per_launch": 2, "stream_depth": 2, "stream_sync": 2},
                                  {"stream_share": 1}, {"steps_per_launch": 2, "stream_share": 1}, {"stream_prefetch": 1}])
def test_row_streaming_kernel_options_do_not_change_results(L, O, opts):
    """Chunk height, rows in flight and the strip-synchronising barriers of kernels_2d_stream.hip only move work
    around: random real data, bit-for-bit the same grid as the default configuration (and the oracle to rounding)."""
    rng = np.random.default_rng(11)
    for shape, dims in (("star2d1r", (301, 1000)), ("box2d3r", (100, 250)), ("star2d3r", (64, 2100))):
        w = O.effective_weights(shape)
        w = w / w.sum()
        a = rng.standard_normal(O.padded_shape(shape, dims))
        opts = dict({"steps_per_launch": 4}, **opts)
        base = plan_run(L, shape, a, 9, weights=w, options={"steps_per_launch": opts["steps_per_launch"]})
        assert np.array_equal(plan_run(L, shape, a, 9, weights=w, options=opts), base), (shape, opts)
        assert rel_err(base, O.run(shape, a, 9, weights=w)) < 1e-13


@pytest.mark.parametrize("opts", [{"wg_rows": 8}, {"wg_rows": 57, "wg_edge_pct": 0}, {"wg_edge_pct": 100}, {"wg_prio": 0},
                                  {"wg_prio": 6, "wg_rows": 100}, {"lowrank_valu": 0}, {"lowrank_valu": 4}])
def test_workgroup_row_kernel_options_do_not_change_results(L, O, opts):
    """Chunk heights (interior and rim strips), the time-sliced wave priorities and the tap evaluation of kernels_2d_wg.hip
    only move work around: random real data, the same grid as the default configuration bit for bit (1e-13 across tap
    evaluations, which sum in another order), and the oracle to rounding."""
    rng = np.random.default_rng(12)
    for shape, dims in (("star2d1r", (301, 1000)), ("box2d3r", (100, 250)), ("star2d3r", (64, 2100)), ("star2d1r", (1300, 600))):
        w = O.effective_weights(shape)
        w = w / w.sum()
        a = rng.standard_normal(O.padded_shape(shape, dims))
        base = plan_run(L, shape, a, 13, weights=w)
        got = plan_run(L, shape, a, 13, weights=w, options=opts)
        if "lowrank_valu" in opts:
            assert rel_err(got, base) < 1e-13, (shape, opts)
        else:
            assert np.array_equal(got, base), (shape, opts)
        assert rel_err(base, O.run(shape, a, 13, weights=w)) < 1e-13


@pytest.mark.parametrize("zc", [1, 2, 4, 7, 16, 40])
def test_3d_z_chunk_does_not_change_results(L, O, zc):
    for shape in ("star3d1r", "box3d1r"):
        dims = (23, 20, 130)
        a = O.reference_input(shape, dims)
        assert np.array_equal(plan_run(L, shape, a, 2, options={"z_chunk": zc}), O.run(shape, a, 2))


@pytest.mark.parametrize("shape,dims,t", [("star2d1r", (64, 128), 30), ("box2d3r", (64, 128), 25),
                                          ("star2d3r", (64, 128), 40), ("star3d1r", (16, 16, 128), 40),
                                          ("box3d1r", (16, 16, 128), 30), ("1d1r", (4096,), 60)])
def test_long_runs_within_relative_tolerance(L, O, shape, dims, t):
    """Past ~6-15 steps the values stop being exact integers: rounding differs between the FMA
    kernels and the oracle's multiply + add, and must stay far inside the 1e-10 relative bound."""
    a = O.reference_input(shape, dims)
    got = plan_run(L, shape, a, t)
    exp = O.run(shape, a, t)
    g_i, e_i = O.interior(shape, got), O.interior(shape, exp)
    assert np.isfinite(e_i).all()
    assert rel_err(g_i, e_i) < REL_TOL
    assert np.abs(g_i / e_i - 1.0).max() < REL_TOL  # point-wise too: all taps are positive here or cancel mildly


@pytest.mark.parametrize("shape,dims", [("star2d1r", (96, 256)), ("star2d3r", (96, 256)), ("box2d3r", (96, 256)),
                                        ("star3d1r", (10, 24, 128)), ("box3d1r", (10, 24, 128)), ("1d1r", (5000,))])
def test_random_real_weights_and_inputs(L, O, shape, dims):
    """Explicit taps (normalised-weights mode, lora_plan_set_weights): gaussian inputs and weights."""
    rng = np.random.default_rng(123)
    a = rng.standard_normal(O.padded_shape(shape, dims))
    w = rng.standard_normal(O.NTAPS[len(dims)])
    if shape == "star2d1r":  # keep the 25-tap diamond support
        m = np.add.outer(np.abs(np.arange(7) - 3), np.abs(np.arange(7) - 3)) <= 3
        w = (w.reshape(7, 7) * m).ravel()
    if shape in ("star2d3r",):
        m = np.zeros((7, 7), bool)
        m[3, :] = m[:, 3] = True
        w = (w.reshape(7, 7) * m).ravel()
    if shape == "star3d1r":
        m = np.zeros((3, 3, 3), bool)
        m[1, 1, :] = m[1, :, 1] = m[:, 1, 1] = True
        w = (w.reshape(3, 3, 3) * m).ravel()
    w /= np.abs(w).sum()
    for t in (1, 5):
        got = plan_run(L, shape, a, t, weights=w)
        exp = O.run(shape, a, t, weights=w)
        if a.ndim == 1:
            exp[-1] = got[-1]
        assert rel_err(got, exp) < 1e-13, f"{shape} t={t}"


# ---------------------------------------------------------------------------------------------------------
# bf16 storage (BASELINE config 5; new capability -- the oracle for it is unpinned by the reference)
# ---------------------------------------------------------------------------------------------------------
def plan_run_bf16(L, shape, bits, times, weights=None, options=None):
    import torch

    dims = (bits.shape[0] - 2, bits.shape[1] - 4, bits.shape[2] - 8)
    plan = L.Plan(shape, dims, dtype="bf16")
    if weights is not None:
        plan.set_weights(weights)
    for k, v in (options or {}).items():
        plan.set_option(k, v)
    b0 = torch.from_numpy(bits.view(np.int16).copy()).cuda().view(torch.bfloat16)
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, times)
    torch.cuda.synchronize()
    return (b0, b1)[times % 2].view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("shape", ["box3d1r", "star3d1r"])
@pytest.mark.parametrize("dims", [(8, 16, 128), (9, 21, 264), (5, 3, 8), (20, 40, 520)])
def test_bf16_sweeps_match_oracle_bit_exact(L, O, shape, dims):
    """Same taps (fp32), same fused multiply-add order, one rounding to bf16 per sweep: identical bit patterns."""
    bits = O.to_bf16(O.reference_input(shape, dims))
    for t in (1, 2, 5):
        assert np.array_equal(plan_run_bf16(L, shape, bits, t), O.run_bf16(shape, bits, t)), f"{shape} {dims} t={t}"
    # normalised taps (the reference taps overflow bf16 at step ~23), all z-chunk lengths
    w = O.effective_weights(shape)
    w = w / w.sum()
    exp = O.run_bf16(shape, bits, 12, weights=w)
    assert np.isfinite(O.from_bf16(exp)).all()
    for zc in (1, 4, 16):
        assert np.array_equal(plan_run_bf16(L, shape, bits, 12, weights=w, options={"z_chunk": zc}), exp)
        # LDS-DMA ring variant (global_load_lds, two planes ahead, hand-counted vmcnt)
        assert np.array_equal(plan_run_bf16(L, shape, bits, 12, weights=w, options={"z_chunk": zc, "lds_dma": 1}), exp)
    for t in (1, 3):
        assert np.array_equal(plan_run_bf16(L, shape, bits, t, options={"lds_dma": 1}), O.run_bf16(shape, bits, t))
        # 8 columns per lane (1 KiB row pieces per wave)
        assert np.array_equal(plan_run_bf16(L, shape, bits, t, options={"cols_per_lane": 8}), O.run_bf16(shape, bits, t))


def test_bf16_separable_and_tap_order_forms(L, O):
    """Exactly separable taps (every box3d1r of the reference's API) run as x/y/z passes -- a different fp32
    summation order, restated by oracle_step_3d_bf16_sep; option separable=0 keeps the 27-tap order.  Both must
    equal their oracle bit for bit, make the same separable/not decision as the oracle, and stay within bf16
    precision of each other and of the fp64 result."""
    rng = np.random.default_rng(23)
    shape, dims = "box3d1r", (10, 19, 136)
    bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
    ref_w = O.effective_weights(shape) / 36.0
    aniso = np.einsum("k,i,j->kij", [0.25, 0.5, 0.125], [1.0, 2.0, 1.0], [0.0625, 0.125, 0.03125]).ravel()
    a, b, c = rng.standard_normal((3, 3))
    inexact = np.einsum("k,i,j->kij", a, b, c).ravel() / 8.0       # rank-1 in fp64, not exactly so in fp32
    assert O.separable_27(ref_w) is not None and O.separable_27(aniso) is not None
    for w, name in ((ref_w, "reference"), (aniso, "anisotropic"), (inexact, "inexact")):
        sep = O.separable_27(w) is not None
        eng = L.separable_3x3x3(w)
        assert (eng is not None) == sep, name
        plan = L.Plan(shape, dims, dtype="bf16").set_weights(w)
        assert plan.get_option("tapset") == (2 if sep else 1), name
        plan.close()
        for t in (1, 6):
            exp_sep = O.run_bf16(shape, bits, t, weights=w, separable=True)
            exp_tap = O.run_bf16(shape, bits, t, weights=w, separable=False)
            for opts in ({}, {"lds_dma": 1}, {"cols_per_lane": 8}, {"z_chunk": 3}):
                assert np.array_equal(plan_run_bf16(L, shape, bits, t, weights=w, options=opts), exp_sep), (name, t, opts)
                off = dict(opts, separable=0)
                assert np.array_equal(plan_run_bf16(L, shape, bits, t, weights=w, options=off), exp_tap), (name, t, off)
            if not sep:
                assert np.array_equal(exp_sep, exp_tap)
        # one sweep of either order is the correctly rounded fp64 sum up to fp32 accumulation error: they differ
        # from the exact result by at most one bf16 ulp (2^-8 relative) plus ~27 fp32 roundings of the |taps| sum
        exact = O.run(shape, O.from_bf16(bits), 1, weights=w)
        scale = O.run(shape, np.abs(O.from_bf16(bits)), 1, weights=np.abs(w))
        for form in (True, False):
            got = O.from_bf16(plan_run_bf16(L, shape, bits, 1, weights=w, options={"separable": int(form)}))
            err = np.abs(O.interior(shape, got) - O.interior(shape, exact))
            bound = 2.0 ** -8 * np.abs(O.interior(shape, exact)) + 30 * 2.0 ** -24 * O.interior(shape, scale)
            assert (err <= bound).all(), (name, form)


@pytest.mark.parametrize("shape", ["box3d1r", "star3d1r"])
@pytest.mark.parametrize("dims", [(40, 61, 128), (9, 31, 248), (3, 5, 8), (37, 64, 360)])
def test_bf16_fused_two_step_launches_equal_step_by_step(L, O, shape, dims):
    """stencil3d_bf16_fused2_kernel: level 1 is rounded to bf16 exactly as a single sweep stores it, so fused runs
    (default) equal the step-by-step oracle bit for bit -- whole padded buffer, halo state included."""
    import torch

    rng = np.random.default_rng(31)
    bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
    w = O.effective_weights(shape)
    w = w / w.sum()
    assert L.Plan(shape, dims, dtype="bf16").kernel_name == "stencil3d_bf16_fused2_kernel"
    for t in (4, 5, 7, 8):
        exp = O.run_bf16(shape, bits, t, weights=w)
        for opts in ({}, {"fused_z_chunk": 1}, {"fused_z_chunk": 5}, {"separable": 0}, {"fused_pipeline": 1},
                     {"fused_pipeline": 1, "fused_z_chunk": 2}):
            e = exp if "separable" not in opts else O.run_bf16(shape, bits, t, weights=w, separable=False)
            assert np.array_equal(plan_run_bf16(L, shape, bits, t, weights=w, options=opts), e), (shape, dims, t, opts)
        single = plan_run_bf16(L, shape, bits, t, weights=w, options={"steps_per_launch": 1})
        assert np.array_equal(single, exp)
    # general (non-separable) taps and direct step2 calls on plane ranges
    wr = rng.standard_normal(27)
    wr /= np.abs(wr).sum()
    if shape == "star3d1r":
        wr = np.where(O.effective_weights(shape) != 0, wr, 0.0)
    exp = O.run_bf16(shape, bits, 2, weights=wr)
    plan = L.Plan(shape, dims, dtype="bf16").set_weights(wr)
    src = torch.from_numpy(bits.view(np.int16).copy()).cuda().view(torch.bfloat16)
    dst = src.clone()
    dst[1:-1, 2:-2, 4:-4] = -1.0
    h = dims[0]
    for b, e in ((h // 2, h), (0, h // 3), (h // 3, h // 2)):
        plan.step2_region(src, dst, b, e)
    torch.cuda.synchronize()
    assert np.array_equal(dst.view(torch.int16).cpu().numpy().view(np.uint16), exp)


@pytest.mark.parametrize("dims", [(3, 5, 8), (9, 31, 248), (40, 61, 128), (37, 64, 360), (12, 57, 968), (30, 113, 240), (64, 7, 16)])
def test_bf16_register_resident_kernel_equals_step_by_step(L, O, dims):
    """stencil3d_bf16_lanes_kernel (four applications per launch, the levels in registers, every level rounded to bf16 as a
    single sweep stores it): runs of 4 .. 13 sweeps -- full launches, a two-application tail, single sweeps -- equal the
    step-by-step bf16 oracle bit for bit on the whole padded buffer; tiny, ragged, rim-only and several-tile grids, every
    z-chunk length; general separable taps; taps that are not separable keep the tile kernels."""
    shape = "box3d1r"
    rng = np.random.default_rng(41)
    bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
    w = O.effective_weights(shape)
    w = w / w.sum()
    plan = L.Plan(shape, dims, dtype="bf16").set_weights(w).set_option("steps_per_launch", 4)
    assert plan.kernel_name == "stencil3d_bf16_lanes_kernel" and plan.get_option("steps_per_launch") == 4
    for t in (4, 5, 8, 9, 13):
        exp = O.run_bf16(shape, bits, t, weights=w)
        for opts in ({"steps_per_launch": 4}, {"steps_per_launch": 4, "fused_z_chunk": 3}, {"lanes3": 1, "fused_z_chunk": 16},
                     {"steps_per_launch": 4, "spans3": 1}, {"steps_per_launch": 4, "spans3": 0},
                     {"steps_per_launch": 4, "spans3": 2}):  # (spans.h: spans, chunks, team spans)
            assert np.array_equal(plan_run_bf16(L, shape, bits, t, weights=w, options=opts), exp), (dims, t, opts)
    # any exactly separable taps: a (x) b (x) c of random factors
    a, b, c = (rng.standard_normal(3).astype(np.float32) for _ in range(3))
    ws = (a[:, None, None] * b[None, :, None]).astype(np.float32)[..., None] * c[None, None, None, :]
    ws = np.asarray(ws, dtype=np.float32).reshape(27).astype(np.float64) / 8.0
    sp = L.Plan(shape, dims, dtype="bf16").set_weights(ws).set_option("steps_per_launch", 4)
    if sp.kernel_name == "stencil3d_bf16_lanes_kernel":  # (the exact rank-1 test may refuse products that rounded)
        assert np.array_equal(plan_run_bf16(L, shape, bits, 9, weights=ws, options={"steps_per_launch": 4}),
                              O.run_bf16(shape, bits, 9, weights=ws))
    wr = rng.standard_normal(27)
    wr /= np.abs(wr).sum()
    general = L.Plan(shape, dims, dtype="bf16").set_weights(wr).set_option("steps_per_launch", 4)
    assert general.kernel_name == "stencil3d_bf16_fused2_kernel" and general.get_option("steps_per_launch") == 2
    assert L.Plan("star3d1r", dims, dtype="bf16").set_option("steps_per_launch", 4).kernel_name == "stencil3d_bf16_fused2_kernel"


def test_bf16_register_resident_kernel_two_ranges_in_one_launch(L, O):
    """lora_plan_stepn_region2 on a bf16 grid: the two end regions in ONE launch of the register-resident kernel + the
    interior == one launch over all planes, bit for bit; four and two applications."""
    import torch

    shape, dims = "box3d1r", (45, 70, 248)
    rng = np.random.default_rng(77)
    w = O.effective_weights(shape)
    w = w / w.sum()
    bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
    plan = L.Plan(shape, dims, dtype="bf16").set_weights(w).set_option("steps_per_launch", 4)
    assert plan.kernel_name == "stencil3d_bf16_lanes_kernel"
    src = torch.from_numpy(bits.view(np.int16).copy()).cuda().view(torch.bfloat16)
    for napps in (4, 2):
        whole = src.clone()
        plan.stepn_region(napps, src, whole, 0, dims[0])
        for (b0, e0, b1, e1), rest in (((0, 4, 41, 45), (4, 41)), ((30, 45, 0, 9), (9, 30))):
            dst = src.clone()
            plan.stepn_region2(napps, src, dst, b0, e0, b1, e1)
            plan.stepn_region(napps, src, dst, *rest)
            torch.cuda.synchronize()
            assert torch.equal(dst.view(torch.int16), whole.view(torch.int16)), (napps, b0, e0, b1, e1)


@pytest.mark.parametrize("dims", [(8, 28, 64), (9, 31, 248), (5, 3, 8), (37, 64, 360), (40, 61, 128)])
def test_bf16_matrix_pipe_variant_matches_its_contract(L, O, dims):
    """LORA_VARIANT_MFMA of a bf16 box plan (kernels_3d_bf16_mfma.hip: v_mfma_f32_16x16x32_bf16, two applications per
    launch) against the oracle's restatement of ITS contract (exact 27-term sum -> fp32 -> one scaling -> bf16):
    bit for bit in the exact regime (small integers, normalised taps, one pair of fused launches), within one bf16 ulp
    on gaussian data and over longer runs (tails are single sweeps under the same contract)."""
    import torch

    shape = "box3d1r"
    w0 = O.effective_weights(shape)
    rng = np.random.default_rng(3)
    for wname, w in (("ref", w0), ("norm", w0 / w0.sum())):
        for dname in ("int", "gauss"):
            ps = O.padded_shape(shape, dims)
            a = rng.integers(0, 100, ps).astype(np.float64) if dname == "int" else rng.standard_normal(ps)
            bits = O.to_bf16(a)
            plan = L.Plan(shape, dims, dtype="bf16").set_weights(w)
            plan.set_variant(L.VARIANT_MFMA)
            assert plan.kernel_name == "stencil3d_bf16_mfma2_kernel" and plan.get_option("variant") == L.VARIANT_MFMA
            if dname == "int":
                # ONE fused launch on small integers: every partial sum of both levels is exact in fp32 and the hi + lo
                # split loses nothing -> the contract bit for bit (two sweeps of the oracle, level-1 halo = 0)
                src = torch.from_numpy(bits.view(np.int16)).cuda()
                dst = src.clone()
                plan.step2(src, dst)
                torch.cuda.synchronize()
                assert np.array_equal(dst.cpu().numpy().view(np.uint16),
                                      O.run_bf16(shape, bits, 2, weights=w, separable="mfma")), (dims, wname)
            for times in (4, 5, 9):
                if wname == "ref" and times > 4:
                    continue  # x 36 per sweep: out of the bf16 range soon
                b0 = torch.from_numpy(bits.view(np.int16)).cuda()
                b1 = torch.zeros_like(b0)
                plan.run(b0, b1, times)
                torch.cuda.synchronize()
                got = (b0, b1)[times % 2].cpu().numpy().view(np.uint16)
                exp = O.run_bf16(shape, bits, times, weights=w, separable="mfma")
                g, e = O.from_bf16(got), O.from_bf16(exp)
                assert np.abs(g - e).max() <= 2.0 ** -7 * np.abs(e).max(), (dims, wname, dname, times)
                assert (got != exp).mean() < 0.02  # and nearly everywhere identical
    # the variant exists for box taps with bf16-exact normalised factors only
    with pytest.raises(L.LoraError):
        L.Plan("star3d1r", (8, 16, 64), dtype="bf16").set_variant(L.VARIANT_MFMA)
    with pytest.raises(L.LoraError):
        L.Plan(shape, (8, 16, 64), dtype="bf16").set_weights(rng.standard_normal(27)).set_variant(L.VARIANT_MFMA)
    with pytest.raises(L.LoraError):
        L.Plan(shape, (8, 16, 64)).set_variant(L.VARIANT_MFMA)  # fp64 3D: no matrix-pipe form
    # and the default stays the vector kernel (measured faster: DESIGN.md section 3.4c)
    assert L.Plan(shape, (8, 16, 64), dtype="bf16").kernel_name == "stencil3d_bf16_fused2_kernel"


def test_bf16_host_operator_and_random_taps(L, O):
    rng = np.random.default_rng(11)
    shape, dims = "box3d1r", (6, 10, 64)
    bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
    out, info = L.run_host(shape, bits, times=3)
    assert out.dtype == np.uint16 and np.array_equal(out, O.run_bf16(shape, bits, 3))
    w = rng.standard_normal(27)
    w /= np.abs(w).sum()
    assert np.array_equal(plan_run_bf16(L, shape, bits, 4, weights=w), O.run_bf16(shape, bits, 4, weights=w))
    # the bf16 result tracks the fp64 one to bf16 precision
    a = O.from_bf16(bits)
    ref = O.run(shape, a, 3, weights=O.effective_weights(shape) / 36.0)
    got = O.from_bf16(plan_run_bf16(L, shape, bits, 3, weights=O.effective_weights(shape) / 36.0))
    # (against the largest value: three sweeps of signed data in bf16 carry an error of ~3 bf16 ulps of their ADDENDS)
    gi, ri = O.interior(shape, got), O.interior(shape, ref)
    assert np.abs(gi - ri).max() / np.abs(ri).max() < 3 * 2.0 ** -8


def test_params_are_honoured_like_the_reference(L, O):
    rng = np.random.default_rng(5)
    # star2d1r / star3d1r ignore params; star2d3r reads the centre row and column only; box3d1r reads params[0..2]
    for shape, dims in (("star2d1r", (32, 64)), ("star2d3r", (32, 64)), ("star3d1r", (4, 8, 16)), ("box3d1r", (4, 8, 16)),
                        ("1d1r", (256,))):
        p = np.round(rng.uniform(-4, 4, O.NTAPS[len(dims)]))
        a = O.reference_input(shape, dims)
        got = plan_run(L, shape, a, 2, params=p)
        exp = O.run(shape, a, 2, params=p)
        if a.ndim == 1:
            exp[-1] = got[-1]
        assert np.array_equal(got, exp), shape
    # box2d: a symmetric table whose 4th pyramid term does not vanish -> the rank-3 part is applied
    p = O.default_params("box2d3r").copy()
    p[24] = 10.0
    a = O.reference_input("box2d3r", (32, 64))
    assert np.array_equal(plan_run(L, "box2d3r", a, 2, params=p), O.run("box2d3r", a, 2, params=p))


@pytest.mark.parametrize("shape,dims,t", [("1d1r", (1 << 16,), 40), ("star2d1r", (128, 256), 21), ("star3d1r", (8, 16, 64), 17)])
def test_graph_replay_of_launch_bound_runs(L, O, shape, dims, t):
    """lora_plan_run captures small, many-step runs into a hipGraph (needs a non-default stream) and replays it."""
    import torch

    a = O.reference_input(shape, dims)
    w = O.effective_weights(shape)
    w = w / w.sum()
    exp = O.run(shape, a, t, weights=w)
    plan = L.Plan(shape, dims).set_weights(w)
    plan.set_option("graph", 1)
    stream = torch.cuda.Stream()
    for rep in range(3):  # capture once, then two replays of the cached graph
        b0 = torch.from_numpy(a).cuda()
        b1 = torch.zeros_like(b0)
        if rep:
            keep0.copy_(b0)
            keep1.zero_()
            b0, b1 = keep0, keep1
        else:
            keep0, keep1 = b0, b1
        torch.cuda.synchronize()
        plan.run(b0, b1, t, stream=stream)
        stream.synchronize()
        got = (b0, b1)[t % 2].cpu().numpy()
        if a.ndim == 1:
            got[-1] = exp[-1]
        assert rel_err(got, exp) < 1e-13, rep
    # new taps invalidate the cached graph
    plan.set_weights(w * 0.5)
    keep0.copy_(torch.from_numpy(a))
    keep1.zero_()
    torch.cuda.synchronize()
    plan.run(keep0, keep1, t, stream=stream)
    stream.synchronize()
    got = (keep0, keep1)[t % 2].cpu().numpy()
    exp2 = O.run(shape, a, t, weights=w * 0.5)
    if a.ndim == 1:
        got[-1] = exp2[-1]
    assert rel_err(got, exp2) < 1e-13


@pytest.mark.parametrize("bc", ["dirichlet", "periodic"])
@pytest.mark.parametrize("shape,dims", [("star2d1r", (64, 128)), ("star2d1r", (53, 246)), ("box2d3r", (40, 130)),
                                        ("star2d3r", (200, 380)), ("star3d1r", (9, 20, 136)), ("box3d1r", (6, 5, 8)),
                                        ("1d1r", (4096,)), ("star2d1r", (33, 65)), ("star3d1r", (40, 70, 200)),
                                        ("box3d1r", (33, 35, 130)), ("star2d1r", (300, 1000)), ("box2d3r", (90, 1430)),
                                        ("star2d3r", (13, 952))])
def test_boundary_condition_options(L, O, shape, dims, bc):
    """fixed (Dirichlet) and periodic halos as driver options, against the oracle's restatement."""
    import torch

    a = O.reference_input(shape, dims)
    for t in (1, 4, 5, 8, 9, 14):  # 2D Dirichlet: launches of four (workgroup-row kernel), tails of two, single sweeps
        plan = L.Plan(shape, dims).set_boundary(bc)
        b0 = torch.from_numpy(a).cuda()
        b1 = torch.zeros_like(b0)
        plan.run(b0, b1, t)
        torch.cuda.synchronize()
        got = (b0, b1)[t % 2].cpu().numpy()
        exp = O.run_bc(shape, a, t, bc)
        if np.abs(exp).max() < 2.0 ** 50:
            assert np.array_equal(got, exp), f"{shape} {dims} {bc} t={t}"
        else:
            assert rel_err(got, exp) < 1e-13


@pytest.mark.parametrize("shape,dims,dtype", [("star2d1r", (64, 128), "f64"), ("star2d1r", (53, 247), "f64"), ("box2d3r", (40, 130), "f64"),
                                              ("star2d3r", (200, 380), "f64"), ("star3d1r", (9, 20, 136), "f64"),
                                              ("box3d1r", (33, 35, 130), "f64"), ("star3d1r", (40, 70, 200), "f64"),
                                              ("1d1r", (4096,), "f64"), ("box3d1r", (21, 37, 136), "bf16"),
                                              ("star2d1r", (300, 1000), "f64")])
def test_periodic_runs_in_fused_launches_on_a_ghost_extended_grid(L, O, shape, dims, dtype):
    """The torus by ghost zones (capi.cpp: run_torus): a periodic run goes through the ordinary fused kernels on a grid
    extended by radius x applications cells of periodic images on every side, re-wrapped after every launch.  Equal to the
    oracle's wrap-then-sweep restatement and to the single-sweep path (option torus = 0: to rounding in fp64, bit for bit
    in bf16; test_boundary_condition_options holds the exact-integer runs); the launches are fused ones; the result's halo
    is the periodic image of its interior."""
    import torch

    rng = np.random.default_rng(len(dims) * 100 + dims[0])
    ps = O.padded_shape(shape, dims)
    w = O.effective_weights(shape)
    w = w / w.sum()
    if dtype == "bf16":
        a = O.from_bf16(O.to_bf16(rng.standard_normal(ps)))
        tdt = torch.bfloat16
    else:
        a = rng.standard_normal(ps)
        tdt = torch.float64

    def run(t, torus):
        plan = L.Plan(shape, dims, dtype=dtype).set_weights(w).set_boundary("periodic").set_option("torus", torus)
        b0 = torch.from_numpy(a).to(tdt).cuda()
        b1 = torch.zeros_like(b0)
        prof = plan.run_profiled(b0, b1, t)
        torch.cuda.synchronize()
        return (b0, b1)[t % 2].double().cpu().numpy(), prof

    for t in (2, 6, 7, 13, 20):
        got, prof = run(t, 1)
        single, prof1 = run(t, 0)
        assert prof1.fused_launches == 0 and prof1.single_launches == t
        assert prof.fused_launches + prof.two_launches > 0, (shape, dims, t)  # (t = 2: one two-application launch)
        assert prof.fused_launches * prof.apps_per_fused_launch <= t
        if dtype == "f64":
            # (real data: the fused 2D kernels and the separable 3D box sum in another order than the single sweep's tap loop)
            assert rel_err(got, single) < 1e-12, (shape, dims, t)
        else:
            assert np.array_equal(got, single), (shape, dims, t)  # bf16: every level rounded as a single sweep stores it
        if dtype == "f64":
            assert rel_err(got, O.run_bc(shape, a, t, "periodic", weights=w)) < 1e-12, (shape, dims, t)
        h = L.ops.halo(shape)
        inner = got[tuple(slice(k, -k) for k in h)]
        assert np.array_equal(got, np.pad(inner, [(k, k) for k in h], mode="wrap"))
    # on a real stream a long run of a small grid is captured into a hipGraph and replayed: the extended grid's buffers are
    # there before the capture starts; twice, so that the second run replays
    side = torch.cuda.Stream()
    plan = L.Plan(shape, dims, dtype=dtype).set_weights(w).set_boundary("periodic").set_option("graph", 1)
    for _ in range(2):
        b0 = torch.from_numpy(a).to(tdt).cuda()
        b1 = torch.zeros_like(b0)
        torch.cuda.synchronize()
        plan.run(b0, b1, 20, stream=side)
        side.synchronize()
        assert np.array_equal(b0.double().cpu().numpy(), got), "captured run"  # (got: the 20-sweep run above)
    # grids smaller than a ghost zone keep the single sweeps
    small = {1: (64,), 2: (20, 64), 3: (4, 8, 16)}[len(dims)]
    plan = L.Plan(shape, small, dtype=dtype).set_boundary("periodic")
    b0 = torch.zeros(L.padded_shape(shape, small), dtype=tdt, device="cuda")
    prof = plan.run_profiled(b0, torch.zeros_like(b0), 8)
    assert prof.fused_launches == 0 and prof.single_launches == 8


def test_plan_halo_modes(L, O):
    """lora_plan_halo: copy / zero / wrap of every cell outside the interior, any shape, fp64 and bf16."""
    import torch

    rng = np.random.default_rng(3)
    for shape, dims, dtype in (("1d1r", (37,), "f64"), ("star2d1r", (9, 14), "f64"), ("box3d1r", (5, 6, 16), "f64"),
                               ("box3d1r", (5, 6, 16), "bf16")):
        h = L.ops.halo(shape)
        ps = O.padded_shape(shape, dims)
        a = rng.integers(1, 90, ps).astype(np.float64)
        b = rng.integers(1, 90, ps).astype(np.float64)
        tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
        plan = L.Plan(shape, dims, dtype=dtype)
        inner = tuple(slice(k, -k) for k in h)
        ta, tb = torch.from_numpy(a).to(tdt).cuda(), torch.from_numpy(b).to(tdt).cuda()
        plan.halo(tb, "copy", ta)
        torch.cuda.synchronize()
        exp = a.copy()
        exp[inner] = b[inner]
        assert np.array_equal(tb.double().cpu().numpy(), exp), (shape, "copy")
        plan.halo(tb, "zero")
        torch.cuda.synchronize()
        exp = np.zeros_like(a)
        exp[inner] = b[inner]
        assert np.array_equal(tb.double().cpu().numpy(), exp), (shape, "zero")
        plan.halo(tb, "wrap")
        torch.cuda.synchronize()
        assert np.array_equal(tb.double().cpu().numpy(), np.pad(b[inner], [(k, k) for k in h], mode="wrap")), (shape, "wrap")
    with pytest.raises(L.LoraError):
        L.Plan("star2d1r", (2, 64)).halo(torch.zeros((10, 72), dtype=torch.float64, device="cuda"), "wrap")


def test_boundary_condition_bf16_and_validation(L, O):
    import torch

    # Dirichlet on bf16 grids: the fused launches (level-1 halo = the source's halo) == single sweeps, bit for bit
    rng = np.random.default_rng(41)
    for shape, dims in (("box3d1r", (21, 37, 136)), ("star3d1r", (9, 30, 120))):
        bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
        w = O.effective_weights(shape)
        w = w / w.sum()
        res = []
        for spl in (2, 1):
            plan = L.Plan(shape, dims, dtype="bf16").set_weights(w).set_boundary("dirichlet")
            plan.set_option("steps_per_launch", spl)
            b0 = torch.from_numpy(bits.view(np.int16).copy()).cuda().view(torch.bfloat16)
            b1 = torch.zeros_like(b0)
            plan.run(b0, b1, 9)
            torch.cuda.synchronize()
            res.append(b1.view(torch.int16).cpu().numpy().view(np.uint16))
        assert np.array_equal(res[0], res[1]), shape
        assert np.array_equal(res[0][0], bits[0]) and np.array_equal(res[0][:, :2], bits[:, :2])  # halo = the input's

    shape, dims = "box3d1r", (6, 10, 64)
    bits = O.to_bf16(O.reference_input(shape, dims))

    plan = L.Plan(shape, dims, dtype="bf16").set_boundary("periodic")
    w = O.effective_weights(shape) / 36.0
    plan.set_weights(w)
    b0 = torch.from_numpy(bits.view(np.int16).copy()).cuda().view(torch.bfloat16)
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, 1)
    torch.cuda.synchronize()
    got = O.from_bf16(b1.view(torch.int16).cpu().numpy().view(np.uint16))
    # one periodic sweep in bf16 tracks the fp64 periodic sweep to bf16 precision, halo = periodic image
    ref = O.run_bc(shape, O.from_bf16(bits), 1, "periodic", weights=w)
    assert rel_err(got, ref) < 2.0 ** -7
    assert np.array_equal(got[0, 2:-2, 4:-4], got[-2, 2:-2, 4:-4])
    with pytest.raises(L.LoraError):
        L.Plan("star2d1r", (2, 64)).set_boundary("periodic")  # extent smaller than the halo: no torus


def test_halo_cells_are_never_written(L, O):
    import torch

    shape, dims = "star2d1r", (64, 128)
    a = O.reference_input(shape, dims)
    plan = L.Plan(shape, dims)
    src = torch.from_numpy(a).cuda()
    dst = torch.full_like(src, -7.0)
    plan.step(src, dst)
    torch.cuda.synchronize()
    d = dst.cpu().numpy()
    halo = np.ones(d.shape, bool)
    halo[4:-4, 4:-4] = False
    assert (d[halo] == -7.0).all() and (d[~halo] != -7.0).all()
    shape, dims = "box3d1r", (6, 10, 20)
    a = O.reference_input(shape, dims)
    plan = L.Plan(shape, dims)
    src = torch.from_numpy(a).cuda()
    dst = torch.full_like(src, -7.0)
    plan.step(src, dst)
    torch.cuda.synchronize()
    d = dst.cpu().numpy()
    halo = np.ones(d.shape, bool)
    halo[1:-1, 2:-2, 4:-4] = False
    assert (d[halo] == -7.0).all() and (d[~halo] != -7.0).all()


@pytest.mark.parametrize("shape,dims,cuts", [("star2d1r", (160, 256), (0, 32, 37, 128, 160)),
                                             ("box3d1r", (20, 16, 128), (0, 1, 7, 19, 20)),
                                             ("1d1r", (4096,), (0, 1024, 1030, 4096))])
def test_regions_compose_to_the_full_sweep(L, O, shape, dims, cuts):
    """lora_plan_step_region: boundary strips first, interior later (the slab driver's schedule) = one sweep."""
    import torch

    a = O.reference_input(shape, dims)
    plan = L.Plan(shape, dims)
    src = torch.from_numpy(a).cuda()
    dst = torch.zeros_like(src)
    spans = list(zip(cuts[:-1], cuts[1:]))
    for b, e in spans[::2] + spans[1::2]:  # out of order on purpose
        plan.step_region(src, dst, b, e)
    torch.cuda.synchronize()
    exp = O.step(shape, a, O.effective_weights(shape))
    assert np.array_equal(dst.cpu().numpy(), exp)


def test_slab_driver_single_rank_on_gpu(L, O):
    """world_size 1: SlabDriver + HipStepper reproduce the operator (the N > 1 exchange is covered on CPU/gloo)."""
    from lorastencil_amd import slab

    for shape, dims, t in (("star2d1r", (128, 256), 7), ("box2d3r", (64, 128), 3), ("star3d1r", (9, 16, 64), 4)):
        a = O.reference_input(shape, dims)
        drv = slab.SlabDriver(shape, dims, device="cuda:0")
        assert drv.fused  # every 2D and 3D tap set uses the fused two-application launches
        drv.load_global(a)
        drv.run(3)
        drv.run(t - 3)
        assert np.array_equal(drv.gather_global().numpy(), O.run(shape, a, t)), shape


def test_randomised_parity_against_the_oracle(L, O):
    """tools/fuzz_parity.py with a fixed seed: random shapes, ragged / odd / tiny sizes, step counts, boundary options,
    kernel options and taps; whole padded result vs the oracle (bit-exact for integer data and bf16 grids)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(20261004)
    kinds = {}
    for _ in range(300):
        c = fz.draw_case(rng)
        ok, how = fz.run_case(c, rng)
        kinds[how] = kinds.get(how, 0) + 1
        assert ok, c
    assert kinds.get("exact", 0) > 20 and kinds.get("bits", 0) > 10 and kinds.get("rel", 0) > 50


def test_randomised_sweep_of_the_round_4_launch_paths(L, O):
    """tools/fuzz_r04.py for twenty seconds with a fixed seed (its case sequence is the same every time): the
    register-resident 3D kernels under every cut of a launch, fp64 and bf16, two ranges in one launch, the periodic option on
    a ghost-extended grid, full-rank 49-tap tables -- each against single sweeps of the same plan family.  (The sweep that
    found the missing barrier of the fp64 separable path's one-level steps.)"""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_r04.py"), "--seconds", "20", "--seed", "20261005"],
                       capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and "cases agree" in r.stdout, tail
    assert "MISMATCH" not in r.stdout and "ERROR" not in r.stdout, tail


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties + sampled windows against the oracle
# ---------------------------------------------------------------------------------------------------------
FULL = [("star2d1r", (16384, 16384)), ("box2d3r", (8192, 8192)), ("star3d1r", (512, 512, 512))]


@pytest.mark.parametrize("shape,dims", FULL)
def test_full_size_constant_field_and_windows(L, O, shape, dims):
    import torch

    ps = L.padded_shape(shape, dims)
    w = O.effective_weights(shape)
    wsum = w.sum()
    plan = L.Plan(shape, dims)
    # (1) constant field: ones everywhere (halo included) -> every interior point = sum(w) after one sweep;
    #     after the second sweep (halo of buffer 1 is zero) every point further than 2r from the edge = sum(w)^2
    b0 = torch.ones(ps, dtype=torch.float64, device="cuda")
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, 2)
    torch.cuda.synchronize()
    h = L.ops.halo(shape)
    r = {2: 3, 3: 1}[len(dims)]
    inner1 = tuple(slice(k, s - k) for k, s in zip(h, ps))
    assert bool((b1[inner1] == wsum).all())
    inner2 = tuple(slice(k + r, s - k - r) for k, s in zip(h, ps))
    assert bool((b0[inner2] == wsum * wsum).all())
    assert float(b0[inner1].min()) < wsum * wsum  # the zero halo of buffer 1 was really read near the edge
    del b0, b1
    # (2) random small-integer field generated on the device: sampled windows of one sweep, exact vs the oracle
    gen = torch.Generator(device="cuda").manual_seed(2024)
    src = torch.randint(0, 100, ps, generator=gen, device="cuda").to(torch.float64)
    dst = torch.zeros_like(src)
    plan.step(src, dst)
    torch.cuda.synchronize()
    rng = np.random.default_rng(0)
    win = {2: (70, 140), 3: (6, 20, 140)}[len(dims)]
    corners = [tuple(0 for _ in dims), tuple(d - wd for d, wd in zip(dims, win))]
    corners += [tuple(int(rng.integers(0, d - wd)) for d, wd in zip(dims, win)) for _ in range(6)]
    for c in corners:
        sl = tuple(slice(ci, ci + wd + 2 * k) for ci, wd, k in zip(c, win, h))
        sub = src[sl].cpu().numpy()
        exp = O.interior(shape, O.step(shape, sub, w))
        got = O.interior(shape, dst[sl].cpu().numpy())
        assert np.array_equal(got, exp), f"window at {c}"
    # (3) linearity on exact integers: sweep(2*src) == 2*sweep(src)
    dst2 = torch.zeros_like(src)
    src.mul_(2.0)
    plan.step(src, dst2)
    torch.cuda.synchronize()
    assert bool((dst2 == 2.0 * dst).all())
    # (4) the fused kernels == single sweeps through buffers whose halo alternates between 0 and the input's, everywhere
    #     (small integers: exact whatever the summation order of the low-rank evaluation)
    k_apps = plan.get_option("steps_per_launch")
    assert k_apps == (6 if len(dims) == 2 else (4 if shape == "star3d1r" else 2))
    del dst
    two = src.clone()                 # buffer 0 again after two sweeps: the input's halo, new interior
    plan.step(dst2, two)              # dst2 = sweep(src) with a zero halo
    fused = src.clone()
    plan.step2(src, fused)
    torch.cuda.synchronize()
    assert torch.equal(fused, two)
    if k_apps == 6:                   # the default 2D launch: six applications (and the four of its tail launches)
        three = torch.zeros_like(src)
        plan.step(two, three)
        four = src.clone()
        plan.step(three, four)
        plan.step(four, three)
        six = src.clone()
        plan.step(three, six)
        del three
        fused.copy_(src)
        plan.stepk(src, fused)
        torch.cuda.synchronize()
        if float(six.abs().max()) < 2.0 ** 50:
            assert torch.equal(fused, six)
        else:  # box2d3r: 96 x 256^6 has left the exact-integer range, the structured evaluation rounds in another order
            assert float((fused - six).abs().max()) <= 1e-13 * float(six.abs().max())
        del six
        fused.copy_(src)
        plan.set_option("steps_per_launch", 4)
        plan.stepk(src, fused)
        torch.cuda.synchronize()
        assert torch.equal(fused, four)
        plan.set_option("steps_per_launch", 6)
    if k_apps == 4 and len(dims) == 3:   # the default launch of a big 3D star grid: four applications, levels in registers
        assert plan.kernel_name == "stencil3d_lanes_kernel"
        three = torch.zeros_like(src)
        plan.step(two, three)
        four = src.clone()
        plan.step(three, four)
        fused.copy_(src)
        plan.stepk(src, fused)
        torch.cuda.synchronize()
        assert torch.equal(fused, four)
        del four
        # and the plane-streaming kernel's three (interior; its halo is the output's), which such grids ran before
        plan.set_option("steps_per_launch", 3)
        assert plan.kernel_name == "stencil3d_planes_kernel"
        fused.zero_()
        plan.stepk(src, fused)
        torch.cuda.synchronize()
        assert torch.equal(fused, three)


@pytest.mark.parametrize("shape,dims,dtype", [
    ("star2d1r", (49152, 49152), "f64"),        # 2.4e9 points, 19.3 GB per buffer: element indices exceed 2^31
    ("star3d1r", (1600, 1200, 1280), "f64"),    # 2.5e9 points, 19.7 GB per buffer
    ("box3d1r", (2400, 1280, 1536), "bf16"),    # 4.7e9 points, 9.5 GB per buffer: byte offsets exceed 2^32
    ("1d1r", (2 ** 31 - 4096,), "f64"),         # 17.2 GB per buffer: byte offsets exceed 2^32, indices reach 2^31
])
def test_grids_beyond_32_bit_indexing(L, O, shape, dims, dtype):
    """288 GB of HBM allow grids whose element counts and byte offsets do not fit 32 bits: sampled windows of one
    single sweep and of one fused launch near the far end of the array, exact against the oracle."""
    import torch

    plan = L.Plan(shape, dims, dtype=dtype)
    ps = plan.padded_shape
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
    src = torch.empty(ps, dtype=tdt, device="cuda")
    # value = (linear index mod 97) without materialising a 64-bit index tensor: row pattern + per-row offset
    h = L.ops.halo(shape)
    if len(dims) == 1:
        step = 1 << 24
        for lo in range(0, ps[0], step):
            hi = min(lo + step, ps[0])
            src[lo:hi] = (torch.arange(lo, hi, device="cuda", dtype=torch.int64) % 97).to(tdt)
    else:
        inner = torch.arange(ps[-1], device="cuda", dtype=torch.int64)
        flat = src.view(-1, ps[-1])
        rows = flat.shape[0]
        blk = 1 << 14
        for lo in range(0, rows, blk):
            hi = min(lo + blk, rows)
            r = torch.arange(lo, hi, device="cuda", dtype=torch.int64)[:, None]
            flat[lo:hi] = ((r * 31 + inner[None, :] * 7) % 97).to(tdt)
    dst = torch.zeros_like(src)
    w = O.effective_weights(shape)  # integer taps on integers 0..96: every partial sum is exact, whatever the order
    win = {1: (5000,), 2: (70, 140), 3: (12, 24, 136)}[len(dims)]  # (3D: four sweeps per launch leave a 6 x 18 x 130 core)
    corners = [tuple(0 for _ in dims), tuple(d - wd for d, wd in zip(dims, win)),
               tuple((d - wd) // 2 for d, wd in zip(dims, win))]

    def window(t, c, reach):
        sl = tuple(slice(ci, ci + wd + 2 * k) for ci, wd, k in zip(c, win, h))
        x = t[sl].contiguous()
        return x.view(torch.int16).cpu().numpy().view(np.uint16) if dtype == "bf16" else x.cpu().numpy()

    def interior(x, reach):
        return x[tuple(slice(k + reach, x.shape[i] - k - reach) for i, k in enumerate(h))]

    radius = {1: 4, 2: 3, 3: 1}[len(dims)]
    plan.step(src, dst)
    torch.cuda.synchronize()
    for c in corners:
        sub = window(src, c, 0)
        exp = O.run_bf16(shape, sub, 1, weights=w) if dtype == "bf16" else O.step(shape, sub, w)
        assert np.array_equal(interior(window(dst, c, 0), 0), interior(exp, 0)), f"single sweep, window at {c}"
    # one fused launch (2 applications, 8 in 1D): windows well inside the grid see no boundary effect, so K sweeps of
    # the oracle on the window agree on its core (reach = radius x (K - 1) cells from the window's rim)
    k_apps = plan.get_option("steps_per_launch")
    assert k_apps >= 2
    dst.copy_(src)
    plan.stepk(src, dst)
    torch.cuda.synchronize()
    c = corners[2]
    sub = window(src, c, 0)
    exp = O.run_bf16(shape, sub, k_apps, weights=w) if dtype == "bf16" else O.run(shape, sub, k_apps, weights=w)
    reach = radius * (k_apps - 1)
    got = interior(window(dst, c, 0), reach)
    want = interior(exp, reach)
    assert np.array_equal(got, want)


def test_full_size_bf16_box3d1r_768(L, O):
    """BASELINE.json configs[4] at full size: box3d1r 768^3 in bf16 (constant-field identities that are exact in
    bf16, and sampled windows of a random field compared bit-for-bit with the bf16 oracle)."""
    import torch

    shape, dims = "box3d1r", (768, 768, 768)
    ps = L.padded_shape(shape, dims)
    plan = L.Plan(shape, dims, dtype="bf16")
    b0 = torch.ones(ps, dtype=torch.bfloat16, device="cuda")
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, 2)
    torch.cuda.synchronize()
    assert bool((b1[1:-1, 2:-2, 4:-4] == 36.0).all())        # sum of the taps, exact in bf16
    assert bool((b0[2:-2, 3:-3, 5:-5] == 1296.0).all())      # 36^2 = 1.010001b x 2^10, still exact
    assert float(b0[1:-1, 2:-2, 4:-4].float().min()) < 1296.0  # the zero halo of buffer 1 was read at the edge
    del b0, b1
    gen = torch.Generator(device="cuda").manual_seed(7)
    src = torch.randint(0, 100, ps, generator=gen, device="cuda").to(torch.bfloat16)
    dst = torch.zeros_like(src)
    plan.step(src, dst)
    torch.cuda.synchronize()
    rng = np.random.default_rng(1)
    win = (5, 18, 264)
    corners = [(0, 0, 0), tuple(d - w for d, w in zip(dims, win))]
    corners += [tuple(int(rng.integers(0, d - w)) for d, w in zip(dims, win)) for _ in range(5)]
    w27 = O.effective_weights(shape)
    for c in corners:
        sl = tuple(slice(ci, ci + wd + 2 * k) for ci, wd, k in zip(c, win, (1, 2, 4)))
        sub = src[sl].contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
        exp = O.run_bf16(shape, sub, 1, weights=w27)[1:-1, 2:-2, 4:-4]
        got = dst[sl].contiguous().view(torch.int16).cpu().numpy().view(np.uint16)[1:-1, 2:-2, 4:-4]
        assert np.array_equal(got, exp), f"window at {c}"
    # the fused kernel (default for runs of >= 4 sweeps) == two single sweeps through a zero-halo buffer, everywhere;
    # normalised taps keep the random field finite in bf16
    plan.set_weights(w27 / 36.0)
    assert plan.kernel_name == "stencil3d_bf16_lanes_kernel" and plan.get_option("steps_per_launch") == 4
    dst.zero_()
    plan.step(src, dst)
    two = src.clone()
    plan.step(dst, two)
    fused = src.clone()
    plan.step2(src, fused)           # the two-application launch of the same kernel (tails of a run)
    torch.cuda.synchronize()
    assert torch.equal(fused.view(torch.int16), two.view(torch.int16))
    # four applications per launch (the register-resident kernel, BASELINE config 5's default) == four single sweeps
    # through buffers whose halo alternates between 0 and the input's, everywhere
    plan.step(two, dst)              # dst = level 3 with a zero halo
    four = src.clone()
    plan.step(dst, four)
    fused4 = src.clone()
    plan.stepk(src, fused4)
    torch.cuda.synchronize()
    assert torch.equal(fused4.view(torch.int16), four.view(torch.int16))
    # ... and the round-3 tile kernel (two per launch: option lanes3 = 0) still agrees
    old = L.Plan(shape, dims, dtype="bf16").set_weights(w27 / 36.0).set_option("lanes3", 0)
    assert old.kernel_name == "stencil3d_bf16_fused2_kernel"
    fused.copy_(src)
    old.step2(src, fused)
    torch.cuda.synchronize()
    assert torch.equal(fused.view(torch.int16), two.view(torch.int16))


# ---------------------------------------------------------------------------------------------------------
# the CLIs and the reference's own harness (oracle/_ref) on top of the HIP engine
# ---------------------------------------------------------------------------------------------------------
def _run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("dim,args", [(1, ["1d1r", "4096", "3"]), (1, ["1d2r", "2048", "2"]),
                                      (2, ["star2d1r", "64", "128", "3"]), (2, ["box2d3r", "64", "128", "2"]),
                                      (2, ["box2d1r", "32", "64", "2"]), (2, ["star2d3r", "96", "64", "2"]),
                                      (3, ["star3d1r", "8", "16", "128", "2"]), (3, ["box3d1r", "8", "8", "64", "2"])])
def test_cli_runs_and_self_check_passes(L, dim, args):
    rc, out, err = _run([os.path.join(ROOT, "lorastencil_amd", "bin", f"lorastencil_{dim}d"), *args, "--check"])
    assert rc == 0, out + err
    lines = out.splitlines()
    assert lines[0].startswith("INFO: shape = ")
    assert any(l.startswith("LoRAStencil(") for l in lines) and any(l.startswith("GStencil/s = ") for l in lines)
    assert "Comparing naive and lora" in out and lines[-1] == "Correct!"
    assert not any(l.startswith(("row = ", "col = ", "height = ")) for l in lines)


def test_cli_normalize_keeps_config_3_finite(L):
    """BASELINE config 3 (`box2d3r 8192 8192 200`) overflows fp64 with the reference's integer taps (sum 232: NaN from
    step ~129); --normalize (taps / sum) keeps the same run finite from the reference's own surface.  Run
    here at 1024 x 1024: the horizon depends on the step count, not on the size."""
    exe = os.path.join(ROOT, "lorastencil_amd", "bin", "lorastencil_2d")
    rc, out, err = _run([exe, "box2d3r", "1024", "1024", "200", "--normalize"])
    assert rc == 0, out + err
    rng = [l for l in out.splitlines() if l.startswith("Result range = ")]
    assert "Taps normalised" in out and len(rng) == 1 and rng[0].startswith("Result range = [") and "not finite" not in out
    rc, out_u, err = _run([exe, "box2d3r", "1024", "1024", "200"])
    assert rc == 0 and "Result range = not finite" in out_u  # the reference's taps overflow, like the reference itself
    # through the library: same switch, result finite and equal to the oracle with normalised taps
    from lorastencil_amd import _lib

    _lib.lib().lora_set_default_normalize(1)
    try:
        a = L.reference_input("box2d3r", (96, 128))
        out_n, _ = L.run_host("box2d3r", a, times=200, quiet=True)
    finally:
        _lib.lib().lora_set_default_normalize(0)
    assert np.isfinite(out_n).all()
    out_u, _ = L.run_host("box2d3r", a, times=200, quiet=True)
    assert not np.isfinite(out_u).all()  # the reference's taps: overflowed, like the reference itself


@pytest.mark.parametrize("shape,dims", [("star2d1r", (4096, 4096)), ("box2d3r", (2048, 2048)), ("star3d1r", (256, 256, 256)),
                                        ("box3d1r", (256, 256, 256))])
def test_hundred_steps_on_a_large_grid_match_the_oracle(L, O, shape, dims):
    """One long run on a grid large enough to span many chunks / panels, every XCD's share and the whole default
    schedule (2D: sixteen six-application launches of the workgroup-row kernel + one of four; 3D at 256^3: 25 four-
    application launches of the register-resident kernel for the star and the separable box), normalised taps, default
    options, hipGraph off -- against the oracle (all host threads) at the north star's 1e-10."""
    w = O.effective_weights(shape)
    w = w / w.sum()
    a = O.reference_input(shape, dims) if len(dims) == 2 else \
        np.random.default_rng(5).integers(0, 100, O.padded_shape(shape, dims)).astype(np.float64)
    got = plan_run(L, shape, a, 100, weights=w, options={"graph": 0})
    exp = O.run(shape, a, 100, weights=w, threads=O.max_threads())
    assert np.isfinite(exp).all()
    assert rel_err(got, exp) < REL_TOL
    g_i, e_i = O.interior(shape, got), O.interior(shape, exp)
    assert np.abs(g_i - e_i).max() <= REL_TOL * np.abs(e_i).max()
    h = L.ops.halo(shape)
    edge = tuple(slice(0, k) for k in h)
    assert np.array_equal(got[edge], exp[edge])  # the halo state after an even number of steps: the input's


@pytest.mark.parametrize("dim,args", [(1, ["1d1r", "4096", "3"]), (2, ["star2d1r", "64", "128", "3"]),
                                      (2, ["box2d3r", "32", "64", "2"]), (2, ["star2d3r", "32", "64", "1"]),
                                      (3, ["star3d1r", "8", "16", "128", "2"]), (3, ["box3d1r", "8", "8", "64", "1"])])
def test_reference_harness_self_check_on_this_engine(L, dim, args):
    """oracle/_ref/ref_lorastencil_Nd_check = the reference's OWN main.cu built with -DCHECK_ERROR and linked to this
    repo's gpu_*() shims: its test_cpu comparison (1e-7) must print no mismatching point."""
    exe = os.path.join(ROOT, "oracle", "_ref", f"ref_lorastencil_{dim}d_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built")
    rc, out, err = _run([exe, *args])
    assert rc == 0, out + err
    lines = out.splitlines()
    assert "Comparing naive and lora" in out and lines[-1] == "Correct!"
    i = lines.index("Comparing naive and lora")
    assert lines[i + 1:] == ["Correct!"], "the reference's self-check printed mismatches:\n" + "\n".join(lines[i:i + 10])


# ---------------------------------------------------------------------------------------------------------
# multi-rank slabs with the REAL HIP stepper: three ranks share this one GPU and exchange ghost zones over gloo
# (RCCL itself needs one GPU per rank; everything else of the N > 1 product path runs here)
# ---------------------------------------------------------------------------------------------------------
def _slab_rank(rank, world, port, shape, dims, times, every, dtype, q, boundary="reference"):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lorastencil_amd import slab
        from oracle import oracle as O

        a = O.reference_input(shape, dims)
        drv = slab.SlabDriver(shape, dims, device="cuda:0", exchange_every=every, dtype=dtype, boundary=boundary)
        if dtype == "bf16":
            bits = O.to_bf16(a)
            drv.load_global(torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16))
        else:
            drv.load_global(a)
        drv.run(times // 2)
        drv.run(times - times // 2)
        full = drv.gather_global(0)
        if rank == 0:
            res = full.view(torch.int16).numpy().view(np.uint16) if dtype == "bf16" else full.numpy()
            q.put((res, drv.fused, drv.layout.ghost))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,dims,times,every,dtype", [
    ("star2d1r", (384, 256), 9, 2, "f64"),     # a six-application launch + 2 + 1, 36-row ghost zones
    ("star2d1r", (768, 384), 16, 4, "f64"),    # 72-row ghost zones, refresh every 4 launches
    ("box2d3r", (300, 130), 6, 3, "f64"),
    ("star3d1r", (24, 20, 64), 7, 2, "f64"),
    ("box3d1r", (24, 20, 64), 6, 3, "bf16"),
    ("1d1r", (30000,), 27, 2, "f64"),          # 8-step fused launches: 64-point ghost zones
])
def test_three_rank_slabs_on_one_gpu_equal_single_rank(L, O, shape, dims, times, every, dtype):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slab_rank, args=(r, world, port, shape, dims, times, every, dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, fused, ghost = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a = O.reference_input(shape, dims)
    if dtype == "bf16":
        exp = O.run_bf16(shape, O.to_bf16(a), times)
        assert np.array_equal(got, exp)
    else:
        # N ranks == 1 rank bit-for-bit (same kernels, same per-point arithmetic whatever the decomposition) ...
        assert np.array_equal(got, plan_run(L, shape, a, times))
        # ... and equal to the oracle: exactly while the values are exact integers, to rounding beyond 2^53
        exp = O.run(shape, a, times)
        if np.abs(exp).max() < 2.0 ** 50:
            assert np.array_equal(got, exp)
        else:
            assert rel_err(got, exp) < 1e-13
    assert fused  # 1D: eight applications per launch, 2D: six (workgroup-row kernel), 3D fp64 (small grids) and bf16: two
    assert ghost == {1: 4, 2: 3, 3: 1}[len(dims)] * {1: 8, 2: 6, 3: 2}[len(dims)] * every


@pytest.mark.parametrize("boundary", ["dirichlet", "periodic"])
@pytest.mark.parametrize("shape,dims,times", [("star2d1r", (384, 256), 9), ("star3d1r", (24, 20, 64), 7),
                                              ("1d1r", (30000,), 21)])
def test_three_rank_dirichlet_slabs_on_one_gpu(L, O, shape, dims, times, boundary):
    """The Dirichlet option (fused launches keep the caller's halo) and the periodic one (a ring of slabs, single
    sweeps after a wrap of the unsplit dimensions) across slabs with the real HIP stepper."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slab_rank, args=(r, world, port, shape, dims, times, 2, "f64", q, boundary))
             for r in range(world)]
    for p in procs:
        p.start()
    got, fused, ghost = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = O.run_bc(shape, O.reference_input(shape, dims), times, boundary)
    assert fused == (boundary == "dirichlet")
    if np.abs(exp).max() < 2.0 ** 50:
        assert np.array_equal(got, exp)
    else:
        assert rel_err(got, exp) < 1e-13


def test_rccl_backend_initialises_and_slab_driver_runs_under_it(L, O):
    """One rank over the real "nccl" (= RCCL) backend: process-group creation with device_id, barrier, all-reduce, a
    SlabDriver run inside an initialised group, and the ghost exchange itself -- a periodic ring of one slab exchanges
    its boundary strips with ITSELF through RCCL (two RCCL ranks would need two GPUs)."""
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LORA_ROOT"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", os.environ["LORA_PORT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from lorastencil_amd import slab
from oracle import oracle as O
a = O.reference_input("star2d1r", (128, 256))
drv = slab.SlabDriver("star2d1r", (128, 256), device="cuda:0")
drv.load_global(a); drv.refresh_ghosts(); drv.run(6)
torch.cuda.synchronize(); dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
ok = np.array_equal(drv.gather_global().numpy(), O.run("star2d1r", a, 6)) and float(t.item()) == 1.5
# the ghost exchange itself over RCCL: a periodic ring of ONE slab sends its boundary strips to itself (P2P batch on
# row slices, boundary-first overlap, waits -- then the all-gather form), compared with the oracle's torus
for shape, dims, steps in (("star2d1r", (256, 384), 7), ("star3d1r", (24, 20, 64), 5), ("1d1r", (30000,), 6)):
    a = O.reference_input(shape, dims)
    exp = O.run_bc(shape, a, steps, "periodic")
    for mode in ("p2p", "allgather"):
        os.environ["LORA_SLAB_EXCHANGE"] = mode
        drv = slab.SlabDriver(shape, dims, device="cuda:0", boundary="periodic", ring_of_one=True, exchange_every=2)
        assert drv.up == 0 and drv.down == 0 and drv.layout.ghost > 0 and drv.exchange_mode == mode
        drv.load_global(a); drv.run(steps)
        got = drv.gather_global().numpy()
        same = np.array_equal(got, exp) if np.abs(exp).max() < 2.0 ** 50 else np.abs(got - exp).max() <= 1e-13 * np.abs(exp).max()
        if not same:
            print("RING_MISMATCH", shape, mode)
        ok = ok and same
# the deferred wait over a STREAM-ORDERED backend: non-periodic, fused (six applications per launch), ghost zones
# refreshed every 2 launches, so a launch sweeps its deep interior, waits for the ghost rows in flight, then sweeps its
# rims.  (A ring of one under the reference boundary is a rehearsal of one rank's share, not a physical result: what
# must hold is that the overlapped / deferred schedule equals the plain one bit for bit.)  The driver is built while a
# side stream is current and run from the default stream: everything it enqueues must follow its own stream.
a = O.reference_input("star2d1r", (512, 384))
res = []
for defer, overlap in (("1", True), ("0", False)):
    os.environ["LORA_SLAB_DEFER_WAIT"] = defer
    os.environ["LORA_SLAB_EXCHANGE"] = "p2p"
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        drv = slab.SlabDriver("star2d1r", (512, 384), device="cuda:0", ring_of_one=True, exchange_every=2, overlap=overlap)
    assert drv.fused and drv.apps == 6 and drv.layout.ghost == 36 and drv.defer_wait == (defer == "1")
    drv.load_global(a); drv.refresh_ghosts(); drv.run(23)
    side.synchronize(); torch.cuda.synchronize()
    res.append(drv.result().cpu().numpy().copy())
if not np.array_equal(res[0], res[1]):
    print("DEFER_MISMATCH")
    ok = False
dist.destroy_process_group()
print("RCCL_OK" if ok else "RCCL_MISMATCH")
'''
    import socket
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LORA_ROOT=ROOT, LORA_PORT=str(port))
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
    assert "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
