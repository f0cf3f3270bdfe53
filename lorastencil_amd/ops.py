"""Host-side mirror of the reference's operator interface, on top of the C ABI (include/lorastencil.h).

* ``gpu_1d1r`` ... ``gpu_star_3d1r``: the reference's seven operators (1d_utils.h:45-47, 2d_utils.h:47-51,
  3d_utils.h:44-48) with the same argument order and meaning, on padded numpy host arrays.
* ``Plan``: the device-resident form of the same sweep (caller-owned device buffers and stream) that the
  benchmark and the multi-GPU slab driver use.
* params tables, the low-rank factor precompute and the glibc ``rand()`` fill of the reference harness.

Everything computes in the HIP library; there is no Python/CPU implementation here.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np

from . import _lib
from ._lib import LoraError, RunInfo, check  # noqa: F401  (re-exported)

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
SHAPE_NAMES = {v: k for k, v in SHAPES.items()}
_dp = ctypes.POINTER(ctypes.c_double)


def shape_id(shape) -> int:
    if isinstance(shape, str):
        sid = _lib.lib().lora_shape_from_name(shape.encode())
        if sid < 0:
            raise ValueError(f"unknown shape {shape!r}")
        return sid
    return int(shape)


def ndim(shape) -> int:
    return _lib.lib().lora_shape_ndim(shape_id(shape))


def ntaps(shape) -> int:
    return _lib.lib().lora_shape_ntaps(shape_id(shape))


def gstencil_factor(shape) -> int:
    return _lib.lib().lora_shape_gstencil_factor(shape_id(shape))


def halo(shape) -> tuple:
    return {1: (4,), 2: (4, 4), 3: (1, 2, 4)}[ndim(shape)]


def padded_shape(shape, dims: Sequence[int]) -> tuple:
    h = halo(shape)
    if len(dims) != len(h):
        raise ValueError(f"{shape} takes {len(h)} sizes, got {len(dims)}")
    return tuple(int(d) + 2 * k for d, k in zip(dims, h))


def interior(shape, a):
    """Interior view of a padded numpy array or torch tensor."""
    h = halo(shape)
    return a[tuple(slice(k, a.shape[i] - k) for i, k in enumerate(h))]


def _dims_arg(dims: Sequence[int]):
    d = list(int(x) for x in dims) + [0] * (3 - len(dims))
    return (ctypes.c_int * 3)(*d)


def _p(a: np.ndarray):
    if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("expected a C-contiguous float64 array")
    return a.ctypes.data_as(_dp)


# ---- group C: host helpers -------------------------------------------------------------------------
def default_params(shape) -> np.ndarray:
    """The params table the reference harness passes (1d/main.cu:77-78, 2d/main.cu:139-195, 3d/main.cu:112-125)."""
    sid = shape_id(shape)
    p = np.zeros(49)
    n = _lib.lib().lora_default_params(sid, _p(p))
    if n < 0:
        raise LoraError(n, "lora_default_params")
    return p[:n].copy()


def effective_weights(shape, params=None) -> np.ndarray:
    """The taps the operator really applies for ``params`` (see include/lorastencil.h group A)."""
    sid = shape_id(shape)
    w = np.zeros(49)
    pp = None if params is None else _p(np.ascontiguousarray(params, dtype=np.float64))
    n = _lib.lib().lora_effective_weights(sid, pp, _p(w))
    if n < 0:
        raise LoraError(n, "lora_effective_weights")
    return w[:n].copy()


def factorize_7x7(params):
    """Low-rank factor precompute of the box2d operator (2d/gpu.cu:280-350): returns (u[4,7], v[4,7], residual)."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    if params.size != 49:
        raise ValueError("params must hold 49 values")
    u = np.zeros((4, 7))
    v = np.zeros((4, 7))
    res = ctypes.c_double(0.0)
    check(_lib.lib().lora_factorize_7x7(_p(params), _p(u), _p(v), ctypes.byref(res)), "lora_factorize_7x7")
    return u, v, res.value


def svd_7x7(weights):
    """Rank-revealing factorisation of any 7x7 tap matrix: returns (u[7,7], v[7,7], sigma[7]) with
    weights = sum_k outer(u[k], v[k]) and sigma descending; truncating after r terms leaves spectral error sigma[r]."""
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    if weights.size != 49:
        raise ValueError("weights must hold 49 values")
    u = np.zeros((7, 7))
    v = np.zeros((7, 7))
    sig = np.zeros(7)
    check(_lib.lib().lora_svd_7x7(_p(weights), _p(u), _p(v), _p(sig)), "lora_svd_7x7")
    return u, v, sig


def separable_3x3x3(weights):
    """Exact rank-1 test of 27 taps in fp32 (what bf16 plans evaluate): returns (c, b, a) float32 factors along
    x, y, z when w[dz,dy,dx] == (a[dz] * b[dy]) * c[dx] holds exactly, else None."""
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    if weights.size != 27:
        raise ValueError("weights must hold 27 values")
    cba = np.zeros(9, dtype=np.float32)
    rc = _lib.lib().lora_separable_3x3x3(_p(weights), cba.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc < 0:
        check(rc, "lora_separable_3x3x3")
    return (cba[0:3].copy(), cba[3:6].copy(), cba[6:9].copy()) if rc == 1 else None


class GlibcRand:
    """glibc ``rand()`` stream; seed 1 is what the reference's un-seeded harness draws from."""

    def __init__(self, seed: int = 1):
        self._st = _lib.Rng()
        _lib.lib().lora_rng_seed(ctypes.byref(self._st), seed)

    def next(self) -> int:
        return _lib.lib().lora_rng_next(ctypes.byref(self._st))

    def fill(self, count: int, mod: int, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.empty(count, dtype=np.float64)
        _lib.lib().lora_fill_rand(_p(out), count, mod, ctypes.byref(self._st))
        return out


def reference_input(shape, dims, rng: GlibcRand | None = None) -> np.ndarray:
    """Padded input filled like the reference harness (FILL_RANDOM): rand()%10000 in 1D (n+9 draws),
    rand()%100 over the whole padded array in 2D/3D."""
    rng = rng or GlibcRand()
    ps = padded_shape(shape, dims)
    if len(ps) == 1:
        a = rng.fill(ps[0], 10000)
        rng.next()  # 1d/main.cu:107 draws one value more than the array holds
        return a
    return rng.fill(int(np.prod(ps)), 100).reshape(ps)


BOUNDARIES = {"reference": _lib.BC_REFERENCE, "dirichlet": _lib.BC_DIRICHLET, "periodic": _lib.BC_PERIODIC}
DTYPES = {"f64": _lib.F64, "fp64": _lib.F64, "float64": _lib.F64, "bf16": _lib.BF16, "bfloat16": _lib.BF16}


def dtype_id(dtype) -> int:
    if isinstance(dtype, str):
        return DTYPES[dtype]
    return int(dtype)


def to_bf16(a: np.ndarray) -> np.ndarray:
    """float64 array -> bf16 bit patterns (uint16), round-to-nearest-even."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty(a.shape, dtype=np.uint16)
    _lib.lib().lora_f64_to_bf16(_p(a), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), a.size)
    return out


def from_bf16(a: np.ndarray) -> np.ndarray:
    """bf16 bit patterns (uint16) -> float64 (exact)."""
    a = np.ascontiguousarray(a, dtype=np.uint16)
    out = np.empty(a.shape, dtype=np.float64)
    _lib.lib().lora_bf16_to_f64(a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), _p(out), a.size)
    return out


# ---- group A: the reference's operators on host arrays -------------------------------------------------
def run_host(shape, in_: np.ndarray, params=None, times: int = 1, quiet: bool = True, out: np.ndarray | None = None):
    """Generic host-buffer operator.  Returns (out, RunInfo).  A uint16 input is taken as bf16 bit patterns."""
    sid = shape_id(shape)
    if in_.dtype == np.uint16:
        in_ = np.ascontiguousarray(in_)
        h = halo(sid)
        dims = [in_.shape[i] - 2 * h[i] for i in range(in_.ndim)]
        if out is None:
            out = np.zeros_like(in_)
        pp = None if params is None else _p(np.ascontiguousarray(params, dtype=np.float64))
        info = RunInfo()
        check(_lib.lib().lora_run_host_dtype(sid, _lib.BF16, in_.ctypes.data, out.ctypes.data, pp, int(times),
                                             _dims_arg(dims), int(quiet), ctypes.byref(info)),
              f"lora_run_host_dtype({SHAPE_NAMES.get(sid, sid)}, bf16)")
        return out, info
    in_ = np.ascontiguousarray(in_, dtype=np.float64)
    h = halo(sid)
    if in_.ndim != len(h):
        raise ValueError("input rank does not match the shape")
    dims = [in_.shape[i] - 2 * h[i] for i in range(in_.ndim)]
    if out is None:
        out = np.zeros_like(in_)
    pp = None if params is None else _p(np.ascontiguousarray(params, dtype=np.float64))
    info = RunInfo()
    check(_lib.lib().lora_run_host(sid, _p(in_), _p(out), pp, int(times), _dims_arg(dims), int(quiet),
                                   ctypes.byref(info)), f"lora_run_host({SHAPE_NAMES.get(sid, sid)})")
    return out, info


def _operator(cname: str, nd: int):
    def op(in_, out, params, times, *sizes):
        if len(sizes) != nd:
            raise TypeError(f"{cname[5:]} takes {nd} size argument(s)")
        fn = getattr(_lib.lib(), cname)
        check(fn(_p(in_), _p(out), _p(np.ascontiguousarray(params, dtype=np.float64)), int(times),
                 *[int(s) for s in sizes]), cname)

    op.__name__ = cname[5:]
    op.__doc__ = f"Drop-in for the reference's {cname[5:]}(in, out, params, times, sizes...); prints its three lines."
    return op


gpu_1d1r = _operator("lora_gpu_1d1r", 1)
gpu_1d2r = _operator("lora_gpu_1d2r", 1)
gpu_star_2d1r = _operator("lora_gpu_star_2d1r", 2)
gpu_star_2d3r = _operator("lora_gpu_star_2d3r", 2)
gpu_box_2d3r = _operator("lora_gpu_box_2d3r", 2)
gpu_box_3d1r = _operator("lora_gpu_box_3d1r", 3)
gpu_star_3d1r = _operator("lora_gpu_star_3d1r", 3)


# ---- group B: device-resident plans ---------------------------------------------------------------------
def _ptr(x) -> int:
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    return int(x)


def _stream(stream) -> int:
    if isinstance(stream, int):
        return stream
    if stream is None:
        try:
            import torch

            if torch.cuda.is_available():
                return int(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        return 0
    if hasattr(stream, "cuda_stream"):
        return int(stream.cuda_stream)
    return int(stream)


class Plan:
    """One sweep configuration (shape, interior sizes, taps, kernel variant) on device buffers.

    Buffers are padded device arrays (torch CUDA tensors or raw device pointers); ``stream`` is a
    ``torch.cuda.Stream``, a raw ``hipStream_t`` or None (= torch's current stream).
    """

    def __init__(self, shape, dims: Sequence[int], params=None, dtype="f64"):
        self.shape = shape_id(shape)
        self.dtype = dtype_id(dtype)
        self.dims = tuple(int(d) for d in dims)
        if len(self.dims) != ndim(self.shape):
            raise ValueError("wrong number of sizes for the shape")
        self._h = ctypes.c_void_p()
        pp = None if params is None else _p(np.ascontiguousarray(params, dtype=np.float64))
        check(_lib.lib().lora_plan_create(ctypes.byref(self._h), self.shape, self.dtype, _dims_arg(self.dims), pp),
              "lora_plan_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().lora_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration
    @property
    def padded_shape(self) -> tuple:
        return padded_shape(self.shape, self.dims)

    @property
    def padded_bytes(self) -> int:
        return _lib.lib().lora_plan_padded_bytes(self._h)

    @property
    def kernel_name(self) -> str:
        return _lib.lib().lora_plan_kernel_name(self._h).decode()

    @property
    def kernel_signature(self) -> str:
        """Kernel name + every resolved option that selects the instantiation / launch geometry."""
        return _lib.lib().lora_plan_kernel_signature(self._h).decode()

    @property
    def weights(self) -> np.ndarray:
        w = np.zeros(ntaps(self.shape))
        check(_lib.lib().lora_plan_get_weights(self._h, _p(w), w.size), "lora_plan_get_weights")
        return w

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64).ravel()
        check(_lib.lib().lora_plan_set_weights(self._h, _p(w), w.size), "lora_plan_set_weights")
        return self

    def set_variant(self, variant: int):
        check(_lib.lib().lora_plan_set_variant(self._h, int(variant)), "lora_plan_set_variant")
        return self

    def set_boundary(self, boundary):
        """"reference" (default: halo never written), "dirichlet" (halo fixed) or "periodic" -- applied by run()."""
        b = BOUNDARIES[boundary] if isinstance(boundary, str) else int(boundary)
        check(_lib.lib().lora_plan_set_boundary(self._h, b), "lora_plan_set_boundary")
        return self

    def set_option(self, key: str, value: int):
        check(_lib.lib().lora_plan_set_option(self._h, key.encode(), int(value)), f"lora_plan_set_option({key})")
        return self

    def get_option(self, key: str) -> int:
        v = ctypes.c_int(0)
        check(_lib.lib().lora_plan_get_option(self._h, key.encode(), ctypes.byref(v)), f"lora_plan_get_option({key})")
        return v.value

    @property
    def region_granularity(self) -> int:
        return _lib.lib().lora_plan_region_granularity(self._h)

    # -- execution (asynchronous on the stream)
    def step(self, d_in, d_out, stream=None):
        check(_lib.lib().lora_plan_step(self._h, _ptr(d_in), _ptr(d_out), _stream(stream)), "lora_plan_step")

    def step_region(self, d_in, d_out, begin: int, end: int, stream=None):
        check(_lib.lib().lora_plan_step_region(self._h, _ptr(d_in), _ptr(d_out), int(begin), int(end), _stream(stream)),
              "lora_plan_step_region")

    def step2(self, d_in, d_out, stream=None):
        """Two applications in one launch (d_in must be an even time level; see lora_plan_step2)."""
        check(_lib.lib().lora_plan_step2(self._h, _ptr(d_in), _ptr(d_out), _stream(stream)), "lora_plan_step2")

    def step2_region(self, d_in, d_out, begin: int, end: int, stream=None):
        check(_lib.lib().lora_plan_step2_region(self._h, _ptr(d_in), _ptr(d_out), int(begin), int(end),
                                                _stream(stream)), "lora_plan_step2_region")

    def halo(self, d_dst, mode: str, d_src=None, stream=None):
        """Halo cells of a padded device array: "copy" (from d_src), "zero" or "wrap" (periodic, from d_dst itself)."""
        m = {"copy": 0, "zero": 1, "wrap": 2}[mode]
        check(_lib.lib().lora_plan_halo(self._h, _ptr(d_dst), _ptr(d_src) if d_src is not None else None, m,
                                        _stream(stream)), "lora_plan_halo")

    def stepk(self, d_in, d_out, stream=None):
        """The plan's ``steps_per_launch`` applications in one launch (d_in must be an even time level)."""
        check(_lib.lib().lora_plan_stepk(self._h, _ptr(d_in), _ptr(d_out), _stream(stream)), "lora_plan_stepk")

    def stepk_region(self, d_in, d_out, begin: int, end: int, stream=None):
        check(_lib.lib().lora_plan_stepk_region(self._h, _ptr(d_in), _ptr(d_out), int(begin), int(end),
                                                _stream(stream)), "lora_plan_stepk_region")

    def stepn_region(self, napps: int, d_in, d_out, begin: int, end: int, stream=None):
        """``napps`` applications in one launch: 1, the plan's depth, or a tail depth of its kernel family."""
        check(_lib.lib().lora_plan_stepn_region(self._h, int(napps), _ptr(d_in), _ptr(d_out), int(begin), int(end),
                                                _stream(stream)), "lora_plan_stepn_region")

    def stepn_region2(self, napps: int, d_in, d_out, begin0: int, end0: int, begin1: int, end1: int, stream=None):
        """``napps`` applications over two disjoint ranges: one launch where the kernel family takes two (3D register-resident)."""
        check(_lib.lib().lora_plan_stepn_region2(self._h, int(napps), _ptr(d_in), _ptr(d_out), int(begin0), int(end0), int(begin1),
                                                 int(end1), _stream(stream)), "lora_plan_stepn_region2")

    def run_profiled(self, d_buf0, d_buf1, times: int, stream=None):
        """run() with HIP events around the fused and the single-sweep launches; blocks until the run is done.
        Returns a ``_lib.RunProfile``."""
        prof = _lib.RunProfile()
        check(_lib.lib().lora_plan_run_profiled(self._h, _ptr(d_buf0), _ptr(d_buf1), int(times), _stream(stream),
                                                ctypes.byref(prof)), "lora_plan_run_profiled")
        return prof

    def prepare_run(self, times: int):
        """Allocate now what ``run(..., times)`` would allocate on first need (scratch grid, extended periodic grid)."""
        check(_lib.lib().lora_plan_prepare_run(self._h, int(times)), "lora_plan_prepare_run")
        return self

    def run(self, d_buf0, d_buf1, times: int, stream=None):
        """`times` sweeps ping-ponging from d_buf0; the result is in buffer [times % 2]."""
        check(_lib.lib().lora_plan_run(self._h, _ptr(d_buf0), _ptr(d_buf1), int(times), _stream(stream)),
              "lora_plan_run")


def device_count() -> int:
    return _lib.lib().lora_device_count()
