"""ctypes front-end of the CPU oracle (liblorastencil_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``lorastencil_amd/`` imports it.
See ``lorastencil_oracle.h`` for the reference file:line each function follows.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblorastencil_oracle.so")

SHAPES = {
    "1d1r": 0,
    "1d2r": 1,
    "star2d1r": 2,
    "box2d1r": 3,
    "star2d3r": 4,
    "box2d3r": 5,
    "star3d1r": 6,
    "box3d1r": 7,
}
NDIM = {0: 1, 1: 1, 2: 2, 3: 2, 4: 2, 5: 2, 6: 3, 7: 3}
NTAPS = {1: 9, 2: 49, 3: 27}

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the .so is missing or stale."""
    src = os.path.join(_HERE, "lorastencil_oracle.c")
    hdr = os.path.join(_HERE, "lorastencil_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblorastencil_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_rng_seed.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.oracle_rng_next.argtypes = [ctypes.c_void_p]
        L.oracle_rng_next.restype = ctypes.c_int
        L.oracle_fill_rand.argtypes = [_dp, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        L.oracle_default_params.argtypes = [ctypes.c_int, _dp]
        L.oracle_default_params.restype = ctypes.c_int
        L.oracle_effective_weights.argtypes = [ctypes.c_int, _dp, _dp]
        L.oracle_effective_weights.restype = ctypes.c_int
        L.oracle_factorize_7x7.argtypes = [_dp, _dp, _dp]
        L.oracle_step_1d.argtypes = [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int]
        L.oracle_step_2d.argtypes = [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_step_3d.argtypes = [_dp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_run.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_int]
        L.oracle_run.restype = ctypes.c_int
        L.oracle_run_weights.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_int]
        L.oracle_run_weights.restype = ctypes.c_int
        L.oracle_padded_count.argtypes = [ctypes.c_int, _ip]
        L.oracle_padded_count.restype = ctypes.c_size_t
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_run_bc.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int]
        L.oracle_run_bc.restype = ctypes.c_int
        _u16 = ctypes.POINTER(ctypes.c_uint16)
        L.oracle_f32_to_bf16.argtypes = [ctypes.c_float]
        L.oracle_f32_to_bf16.restype = ctypes.c_uint16
        L.oracle_step_3d_bf16.argtypes = [_u16, _u16, ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int]
        L.oracle_run_bf16.argtypes = [ctypes.c_int, _u16, _u16, _dp, ctypes.c_int, _ip, ctypes.c_int]
        L.oracle_run_bf16.restype = ctypes.c_int
        L.oracle_run_bf16_mode.argtypes = [ctypes.c_int, _u16, _u16, _dp, ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int]
        L.oracle_run_bf16_mode.restype = ctypes.c_int
        _fp = ctypes.POINTER(ctypes.c_float)
        L.oracle_separable_27.argtypes = [_fp, _fp, _fp, _fp]
        L.oracle_separable_27.restype = ctypes.c_int
        L.oracle_step_3d_bf16_sep.argtypes = [_u16, _u16, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def shape_id(shape) -> int:
    return SHAPES[shape] if isinstance(shape, str) else int(shape)


def padded_shape(shape, dims):
    nd = NDIM[shape_id(shape)]
    dims = tuple(int(d) for d in dims)
    assert len(dims) == nd
    if nd == 1:
        return (dims[0] + 8,)
    if nd == 2:
        return (dims[0] + 8, dims[1] + 8)
    return (dims[0] + 2, dims[1] + 4, dims[2] + 8)


def halo(shape):
    nd = NDIM[shape_id(shape)]
    return {1: (4,), 2: (4, 4), 3: (1, 2, 4)}[nd]


def interior(shape, a: np.ndarray) -> np.ndarray:
    """View of the interior points of a padded array."""
    h = halo(shape)
    return a[tuple(slice(k, a.shape[i] - k) for i, k in enumerate(h))]


class Rng:
    """glibc ``rand()`` stream (seed 1 unless given)."""

    def __init__(self, seed: int = 1):
        self._buf = ctypes.create_string_buffer(34 * 4 + 8)
        lib().oracle_rng_seed(self._buf, seed)

    def next(self) -> int:
        return lib().oracle_rng_next(self._buf)

    def fill(self, count: int, mod: int) -> np.ndarray:
        out = np.empty(count, dtype=np.float64)
        lib().oracle_fill_rand(_p(out), count, mod, self._buf)
        return out


def reference_input(shape, dims, rng: Rng | None = None) -> np.ndarray:
    """Padded input exactly as the reference harness fills it (FILL_RANDOM, un-seeded rand())."""
    sid = shape_id(shape)
    rng = rng or Rng()
    ps = padded_shape(sid, dims)
    if NDIM[sid] == 1:
        # 1d/main.cu:107 draws cols+1 values; the extra one lands past the allocation
        vals = rng.fill(ps[0] + 1, 10000)
        return vals[: ps[0]].copy()
    return rng.fill(int(np.prod(ps)), 100).reshape(ps)


def default_params(shape) -> np.ndarray:
    sid = shape_id(shape)
    p = np.zeros(49, dtype=np.float64)
    n = lib().oracle_default_params(sid, _p(p))
    return p[:n].copy()


def effective_weights(shape, params=None) -> np.ndarray:
    sid = shape_id(shape)
    params = default_params(sid) if params is None else np.ascontiguousarray(params, dtype=np.float64)
    w = np.zeros(49, dtype=np.float64)
    n = lib().oracle_effective_weights(sid, _p(params), _p(w))
    return w[:n].copy()


def factorize_7x7(params):
    params = np.ascontiguousarray(params, dtype=np.float64)
    u = np.zeros((4, 7))
    v = np.zeros((4, 7))
    lib().oracle_factorize_7x7(_p(params), _p(u), _p(v))
    return u, v


def step(shape, a: np.ndarray, w: np.ndarray, out: np.ndarray | None = None, threads: int = 1) -> np.ndarray:
    """One kernel application on a padded array; only the interior of ``out`` is written."""
    sid = shape_id(shape)
    nd = NDIM[sid]
    a = np.ascontiguousarray(a, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    assert a.ndim == nd and w.size == NTAPS[nd]
    if out is None:
        out = np.zeros_like(a)
    L = lib()
    if nd == 1:
        L.oracle_step_1d(_p(a), _p(out), _p(w), a.shape[0], threads)
    elif nd == 2:
        L.oracle_step_2d(_p(a), _p(out), _p(w), a.shape[0], a.shape[1], threads)
    else:
        L.oracle_step_3d(_p(a), _p(out), _p(w), a.shape[0], a.shape[1], a.shape[2], threads)
    return out


def run(shape, a: np.ndarray, times: int, params=None, weights=None, threads: int = 1,
        out: np.ndarray | None = None) -> np.ndarray:
    """The operator with the reference's driver semantics.  ``a`` is the padded input.

    With ``weights`` the taps are applied as given; otherwise ``params`` (default: the
    reference harness's table) go through ``effective_weights`` like the reference operator.
    """
    sid = shape_id(shape)
    nd = NDIM[sid]
    a = np.ascontiguousarray(a, dtype=np.float64)
    h = halo(sid)
    dims = (ctypes.c_int * 3)(*[a.shape[i] - 2 * h[i] for i in range(nd)], *([0] * (3 - nd)))
    if out is None:
        out = np.zeros_like(a)
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float64)
        rc = lib().oracle_run_weights(sid, _p(a), _p(out), _p(w), times, dims, threads)
    else:
        p = default_params(sid) if params is None else np.ascontiguousarray(params, dtype=np.float64)
        rc = lib().oracle_run(sid, _p(a), _p(out), _p(p), times, dims, threads)
    if rc != 0:
        raise ValueError(f"oracle_run failed (shape={shape}, times={times})")
    return out


def max_threads() -> int:
    return lib().oracle_max_threads()


# ---- bf16 storage (3D shapes; parity unpinned by the reference: it has no reduced-precision path) -------------
def to_bf16(a: np.ndarray) -> np.ndarray:
    """float -> bf16 bit patterns (uint16), round-to-nearest-even (numpy restatement of oracle_f32_to_bf16)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def from_bf16(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def separable_27(weights):
    """Exact fp32 rank-1 test of 27 taps: (c, b, a) float32 factors along x, y, z, or None."""
    w = np.ascontiguousarray(weights, dtype=np.float64).astype(np.float32).ravel()
    assert w.size == 27
    c, b, a = (np.zeros(3, dtype=np.float32) for _ in range(3))
    fp = ctypes.POINTER(ctypes.c_float)
    ok = lib().oracle_separable_27(w.ctypes.data_as(fp), c.ctypes.data_as(fp), b.ctypes.data_as(fp), a.ctypes.data_as(fp))
    return (c, b, a) if ok else None


def run_bf16(shape, a_bits: np.ndarray, times: int, weights=None, threads: int = 1, separable=True) -> np.ndarray:
    """The 3D operator on bf16 bit patterns with the reference's driver semantics.  ``separable`` (the engine's
    default) evaluates exactly separable taps as x/y/z passes; False is the 27-tap order; "mfma" is the contract of the
    engine's matrix-pipe variant (exact 27-term sum, one fp32 rounding, one scaling, one bf16 rounding)."""
    sid = shape_id(shape)
    a_bits = np.ascontiguousarray(a_bits, dtype=np.uint16)
    w = effective_weights(sid) if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    h = halo(sid)
    dims = (ctypes.c_int * 3)(*[a_bits.shape[i] - 2 * h[i] for i in range(3)])
    out = np.zeros_like(a_bits)
    u16 = ctypes.POINTER(ctypes.c_uint16)
    rc = lib().oracle_run_bf16_mode(sid, a_bits.ctypes.data_as(u16), out.ctypes.data_as(u16), _p(w), times, dims,
                                    threads, 2 if separable == "mfma" else (1 if separable else 0))
    if rc != 0:
        raise ValueError("oracle_run_bf16_mode failed")
    return out


BOUNDARIES = {"reference": 0, "dirichlet": 1, "periodic": 2}


def run_bc(shape, a: np.ndarray, times: int, bc, weights=None, threads: int = 1) -> np.ndarray:
    """The driver with a boundary-condition option (not reference behaviour; see lorastencil_oracle.h)."""
    sid = shape_id(shape)
    nd = NDIM[sid]
    a = np.ascontiguousarray(a, dtype=np.float64)
    w = effective_weights(sid) if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    h = halo(sid)
    dims = (ctypes.c_int * 3)(*[a.shape[i] - 2 * h[i] for i in range(nd)], *([0] * (3 - nd)))
    out = np.zeros_like(a)
    b = BOUNDARIES[bc] if isinstance(bc, str) else int(bc)
    if lib().oracle_run_bc(sid, _p(a), _p(out), _p(w), times, dims, b, threads) != 0:
        raise ValueError("oracle_run_bc failed")
    return out
