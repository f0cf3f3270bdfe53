"""ctypes access to the REAL reference host code built by oracle/build_ref.sh (oracle/_ref/libref{1,2,3}d.so).

TEST INFRASTRUCTURE ONLY.  Exposes the reference's own ``test_cpu`` (1d/main.cu:34-40, 2d/main.cu:38-93,
3d/main.cu:33-68) -- the only checker the reference ships -- so that the oracle restatement can be pinned
bit-for-bit, and ``ref_main`` (the reference harness) for conformance runs on a GPU box.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_dp = ctypes.POINTER(ctypes.c_double)

# Itanium-mangled names of the three overloads of test_cpu(double*, double*, double*, int...)
_TEST_CPU = {1: "_Z8test_cpuPdS_S_i", 2: "_Z8test_cpuPdS_S_ii", 3: "_Z8test_cpuPdS_S_iii"}
_REF_MAIN = "_Z8ref_mainiPPc"

_libs: dict[int, ctypes.CDLL] = {}


def available() -> bool:
    return all(os.path.exists(os.path.join(_REF, f"libref{d}d.so")) for d in (1, 2, 3))


def lib(ndim: int) -> ctypes.CDLL:
    if ndim not in _libs:
        _libs[ndim] = ctypes.CDLL(os.path.join(_REF, f"libref{ndim}d.so"))
    return _libs[ndim]


def test_cpu(a: np.ndarray, params: np.ndarray, out: np.ndarray | None = None) -> np.ndarray:
    """One sweep by the reference's test_cpu on a padded array (interior of ``out`` written)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    if out is None:
        out = np.zeros_like(a)
    fn = getattr(lib(a.ndim), _TEST_CPU[a.ndim])
    fn.restype = None
    fn.argtypes = [_dp, _dp, _dp] + [ctypes.c_int] * a.ndim
    fn(a.ctypes.data_as(_dp), out.ctypes.data_as(_dp), params.ctypes.data_as(_dp), *a.shape)
    return out


def run_chain(a: np.ndarray, params: np.ndarray, times: int) -> np.ndarray:
    """`times` sweeps of test_cpu with the reference driver's ping-pong (buf1 starts at zero, halos are
    never written: 2d/gpu.cu:531-554).  Returns buffer [times % 2] (whole padded array)."""
    buf = [np.ascontiguousarray(a, dtype=np.float64).copy(), np.zeros_like(a, dtype=np.float64)]
    for i in range(times):
        test_cpu(buf[i % 2], params, buf[(i + 1) % 2])
    return buf[times % 2]


def ref_main(ndim: int, argv: list[str]) -> int:
    """Run the reference harness (its main()) in-process; needs a GPU for the gpu_*() shims it calls."""
    fn = getattr(lib(ndim), _REF_MAIN)
    fn.restype = ctypes.c_int
    arr = (ctypes.c_char_p * (len(argv) + 1))(*[s.encode() for s in argv], None)
    return fn(len(argv), arr)
