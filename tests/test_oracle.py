"""CPU tests of the oracle itself: pinned to libc rand(), to the golden fixtures generated from the reference's
own test_cpu, and (when oracle/_ref is present) to that test_cpu directly."""
import ctypes

import numpy as np
import pytest
from conftest import ALL_SHAPES, load_golden

from oracle import oracle as o
from oracle import ref


def test_rng_matches_libc_rand():
    libc = ctypes.CDLL("libc.so.6")
    r = o.Rng()
    assert [r.next() for _ in range(5000)] == [libc.rand() for _ in range(5000)]


def test_rng_first_values_known_answer():
    # glibc rand() with the default seed: the first outputs are a well known sequence
    r = o.Rng()
    assert [r.next() for _ in range(3)] == [1804289383, 846930886, 1681692777]


@pytest.mark.parametrize("shape", ALL_SHAPES)
def test_fill_matches_golden_input(shape):
    g = load_golden(shape)
    a = o.reference_input(shape, tuple(g["dims"]))
    assert np.array_equal(a, g["input"])
    assert np.array_equal(o.default_params(shape), g["params"])


@pytest.mark.parametrize("shape", ALL_SHAPES)
@pytest.mark.parametrize("t", [1, 2, 3, 4])
def test_oracle_matches_golden_bit_exact(shape, t):
    g = load_golden(shape)
    out = o.run(shape, g["input"], t)
    exp = g[f"out_t{t}"]
    if out.ndim == 1:  # the operator leaves the last element of `out` untouched (1d/gpu_1r.cu:134)
        assert np.array_equal(out[:-1], exp[:-1])
        assert out[-1] == 0.0
    else:
        assert np.array_equal(out, exp)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("shape,dims", [("1d1r", (96,)), ("star2d1r", (16, 32)), ("box2d3r", (8, 24)),
                                        ("star2d3r", (12, 20)), ("star3d1r", (5, 6, 18)), ("box3d1r", (4, 9, 10))])
def test_oracle_step_equals_reference_test_cpu_random_params(shape, dims):
    rng = np.random.default_rng(42)
    a = rng.standard_normal(o.padded_shape(shape, dims))
    p = rng.standard_normal(o.NTAPS[len(dims)])
    assert np.array_equal(o.interior(shape, ref.test_cpu(a, p)), o.interior(shape, o.step(shape, a, p)))


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("shape", ALL_SHAPES)
def test_oracle_driver_equals_reference_chain(shape):
    g = load_golden(shape)
    for t in (0, 5, 6):
        exp = ref.run_chain(g["input"], g["params"], t)
        out = o.run(shape, g["input"], t)
        n = -1 if out.ndim == 1 else None
        assert np.array_equal(out.ravel()[:n], exp.ravel()[:n])


def test_effective_weights_of_reference_tables():
    # for the reference's own tables the applied taps equal the table
    for s in ALL_SHAPES:
        assert np.array_equal(o.effective_weights(s), o.default_params(s)), s


def test_effective_weights_ignore_what_the_reference_ignores():
    rng = np.random.default_rng(1)
    junk49, junk27 = rng.standard_normal(49), rng.standard_normal(27)
    assert np.array_equal(o.effective_weights("star2d1r", junk49), o.default_params("star2d1r"))
    assert np.array_equal(o.effective_weights("star3d1r", junk27), o.default_params("star3d1r"))
    w = o.effective_weights("box3d1r", junk27).reshape(3, 3, 3)
    assert np.array_equal(w, np.broadcast_to(junk27[:3], (3, 3, 3)))
    w = o.effective_weights("star2d3r", junk49).reshape(7, 7)
    p = junk49.reshape(7, 7)
    assert np.array_equal(w[:, 3], p[:, 3]) and np.array_equal(np.delete(w[3], 3), np.delete(p[3], 3))
    off = np.ones((7, 7), bool)
    off[3, :] = off[:, 3] = False
    assert not w[off].any()


def test_factorizer_reference_table():
    u, v = o.factorize_7x7(o.default_params("box2d3r"))
    assert u[0].tolist() == [1, 2, 3, 4, 3, 2, 1] and v[0].tolist() == [1, 2, 3, 4, 3, 2, 1]
    assert u[1].tolist() == [0, 1, 0, -1, 0, 1, 0] and v[1].tolist() == [0, 1, 0, -1, 0, 1, 0]
    assert u[2].tolist() == [0, 0, -1, -3, -1, 0, 0] and v[2].tolist() == [0, 0, 1, 3, 1, 0, 0]
    assert v[3][3] == 0.0  # the dropped fourth term vanishes for the reference table only


def test_factorizer_drops_fourth_term():
    p = o.default_params("box2d3r").copy()
    p[24] = 10.0  # the loop's own centre value (2d/main.cu:149-166) makes the matrix rank 4
    w = o.effective_weights("box2d3r", p).reshape(7, 7)
    assert w[3, 3] == 8.0  # the operator still applies the rank-3 part only (2d/gpu.cu:358-369)
    u, v = o.factorize_7x7(p)
    assert v[3][3] == 2.0


def test_halo_alternation_and_parity():
    g = load_golden("star2d1r")
    a = g["input"]
    out1 = o.run("star2d1r", a, 1)
    out2 = o.run("star2d1r", a, 2)
    halo = np.ones(a.shape, bool)
    halo[4:-4, 4:-4] = False
    assert not out1[halo].any()                  # odd step count: buffer 1, halo = 0
    assert np.array_equal(out2[halo], a[halo])   # even: buffer 0, halo = the caller's input halo
    assert np.array_equal(o.run("star2d1r", a, 0), a)


def test_threads_do_not_change_results():
    g = load_golden("box3d1r")
    assert np.array_equal(o.run("box3d1r", g["input"], 3, threads=1), o.run("box3d1r", g["input"], 3, threads=4))


def test_boundary_options_of_the_oracle_have_the_properties_they_claim():
    """Dirichlet / periodic are options beyond the reference; nothing of the reference pins them, so the oracle's own
    restatement is checked against what the words mean."""
    rng = np.random.default_rng(4)
    shape, dims = "star2d1r", (12, 20)
    a = rng.integers(0, 50, o.padded_shape(shape, dims)).astype(np.float64)
    w = o.effective_weights(shape)
    halo = np.ones(a.shape, bool)
    halo[4:-4, 4:-4] = False
    # reference mode through the same entry point
    assert np.array_equal(o.run_bc(shape, a, 3, "reference"), o.run(shape, a, 3))
    # Dirichlet: the halo keeps the caller's values at every level; one sweep equals the reference's first sweep
    for t in (1, 2, 3):
        assert np.array_equal(o.run_bc(shape, a, t, "dirichlet")[halo], a[halo])
    assert np.array_equal(o.interior(shape, o.run_bc(shape, a, 1, "dirichlet")), o.interior(shape, o.run(shape, a, 1)))
    d2 = o.step(shape, o.run_bc(shape, a, 1, "dirichlet"), w)
    assert np.array_equal(o.interior(shape, o.run_bc(shape, a, 2, "dirichlet")), o.interior(shape, d2))
    # periodic: translation invariance on the torus, and equality with an explicit numpy wrap
    core = o.interior(shape, a)
    t = 3
    out = o.interior(shape, o.run_bc(shape, a, t, "periodic"))
    shifted = a.copy()
    shifted[4:-4, 4:-4] = np.roll(core, (5, -7), axis=(0, 1))
    out_s = o.interior(shape, o.run_bc(shape, shifted, t, "periodic"))
    assert np.array_equal(out_s, np.roll(out, (5, -7), axis=(0, 1)))
    cur = core.copy()
    w7 = w.reshape(7, 7)
    for _ in range(t):
        nxt = np.zeros_like(cur)
        for dy in range(7):
            for dx in range(7):
                if w7[dy, dx]:
                    nxt += w7[dy, dx] * np.roll(cur, (3 - dy, 3 - dx), axis=(0, 1))
        cur = nxt
    assert np.array_equal(out, cur)
    # 3D and 1D wraps
    for shape, dims in (("box3d1r", (3, 4, 6)), ("1d2r", (16,))):
        a = rng.integers(0, 9, o.padded_shape(shape, dims)).astype(np.float64)
        out = o.run_bc(shape, a, 2, "periodic")
        core = o.interior(shape, out)
        if out.ndim == 1:
            assert np.array_equal(out[:4], core[-4:]) and np.array_equal(out[-4:], core[:4])
        else:
            assert np.array_equal(out[0, 2:-2, 4:-4], core[-1]) and np.array_equal(out[1:-1, 2:-2, :4], core[:, :, -4:])


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_baseline_config0_1d1r_2pow20_100_steps_cpu_path():
    """BASELINE.json configs[0]: `lorastencil_1d 1d1r 1048576 100`, the CPU-runnable plumbing case: the oracle's
    driver against 100 chained sweeps of the reference's own test_cpu, bit for bit (values leave the exact-integer
    range after ~13 steps, so this also pins the rounding of every later step)."""
    n, t = 1 << 20, 100
    a = o.reference_input("1d1r", (n,))
    p = o.default_params("1d1r")
    exp = ref.run_chain(a, p, t)
    got = o.run("1d1r", a, t, threads=o.max_threads())
    assert np.isfinite(exp).all() and np.array_equal(got[:-1], exp[:-1])


def test_bf16_matrix_pipe_contract_is_the_correctly_rounded_sum():
    """oracle_step_3d_bf16_mfma (the contract of the engine's bf16 MFMA variant): on small integers it equals the
    separable fp32 order exactly (every partial sum is exact); on gaussian data the two contracts differ by at most one
    bf16 ulp and agree nearly everywhere -- both are roundings of the same 27-point sum."""
    from oracle import oracle as O

    rng = np.random.default_rng(1)
    shape, dims = "box3d1r", (6, 10, 24)
    w = O.effective_weights(shape)
    bits = O.to_bf16(rng.integers(0, 100, O.padded_shape(shape, dims)).astype(np.float64))
    assert np.array_equal(O.run_bf16(shape, bits, 2, weights=w, separable="mfma"), O.run_bf16(shape, bits, 2, weights=w))
    g = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
    wn = w / w.sum()
    a, b = O.run_bf16(shape, g, 3, weights=wn, separable="mfma"), O.run_bf16(shape, g, 3, weights=wn)
    fa, fb = O.from_bf16(a), O.from_bf16(b)
    assert np.abs(fa - fb).max() <= 2.0 ** -7 * np.abs(fb).max() and (a != b).mean() < 0.2
    with pytest.raises(ValueError):
        O.run_bf16(shape, g, 1, weights=rng.standard_normal(27), separable="mfma")  # no bf16-exact factors
