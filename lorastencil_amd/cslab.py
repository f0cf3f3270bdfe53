"""ctypes mirror of group D of include/lorastencil.h: the C++ multi-GPU slab driver (csrc/slab.cpp).

``SlabSet`` holds the slabs of one decomposition that live in THIS process -- all of them (one process driving N
devices, what the CLIs' ``--gpus N`` does; or N slabs on one device over the loopback exchange: tests on a one-GPU box)
or a single one (one process per GPU with an RCCL communicator).  ``lorastencil_amd.slab.SlabDriver`` is the same
schedule over ``torch.distributed``."""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np

from . import _lib, ops
from ._lib import SLAB_NO_DEFER, SLAB_NO_FUSION, SLAB_NO_OVERLAP, SLAB_OVERLAP, SLAB_RING_OF_ONE, check  # noqa: F401


def loopback_comms(nranks: int):
    arr = (_lib.SlabComm * nranks)()
    check(_lib.lib().lora_slab_comm_loopback(arr, nranks), "lora_slab_comm_loopback")
    return arr


def rccl_comm(nccl_comm) -> "_lib.SlabComm":
    c = _lib.SlabComm()
    check(_lib.lib().lora_slab_comm_rccl(ctypes.byref(c), ctypes.c_void_p(nccl_comm)), "lora_slab_comm_rccl")
    return c


class SlabSet:
    def __init__(self, shape, global_dims: Sequence[int], nranks: int, comms=None, ranks=None, devices=None, dtype="f64",
                 params=None, weights=None, exchange_every: int = 0, boundary="reference", flags: int = 0, options=None):
        self.shape = ops.shape_id(shape)
        self.dtype = ops.dtype_id(dtype)
        self.global_dims = tuple(int(d) for d in global_dims)
        self.nranks = nranks
        self.ranks = list(range(nranks)) if ranks is None else list(ranks)
        self._keep = []
        self._h = []
        lib = _lib.lib()
        for i, r in enumerate(self.ranks):
            d = _lib.SlabDesc()
            d.shape, d.dtype = self.shape, self.dtype
            for k, v in enumerate(self.global_dims):
                d.global_dims[k] = v
            if params is not None:
                p = np.ascontiguousarray(params, dtype=np.float64)
                self._keep.append(p)
                d.params = p.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            if weights is not None:
                w = np.ascontiguousarray(weights, dtype=np.float64)
                self._keep.append(w)
                d.weights = w.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            d.rank, d.nranks = r, nranks
            d.device = 0 if devices is None else devices[i]
            d.exchange_every = exchange_every
            d.boundary = ops.BOUNDARIES[boundary] if isinstance(boundary, str) else int(boundary)
            d.flags = flags
            if options:
                d.options = ",".join(f"{k}={int(v)}" for k, v in options.items()).encode()
            h = ctypes.c_void_p()
            comm = None if comms is None else ctypes.byref(comms[i])
            check(lib.lora_slab_create(ctypes.byref(h), ctypes.byref(d), comm), "lora_slab_create")
            self._h.append(h)
        self._arr = (ctypes.c_void_p * len(self._h))(*[h.value for h in self._h])

    def close(self):
        for h in getattr(self, "_h", []):
            if h.value:
                _lib.lib().lora_slab_destroy(h)
                h.value = None
        self._h = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self, i: int = 0) -> "_lib.SlabInfo":
        si = _lib.SlabInfo()
        check(_lib.lib().lora_slab_info(self._h[i], ctypes.byref(si)), "lora_slab_info")
        return si

    def load(self, global_padded: np.ndarray):
        a = np.ascontiguousarray(global_padded)
        for h in self._h:
            check(_lib.lib().lora_slab_load(h, a.ctypes.data), "lora_slab_load")

    def refresh_ghosts(self):
        check(_lib.lib().lora_slab_refresh_ghosts_many(self._arr, len(self._h)), "lora_slab_refresh_ghosts_many")

    def run(self, times: int):
        check(_lib.lib().lora_slab_run_many(self._arr, len(self._h), int(times)), "lora_slab_run_many")

    def sync(self):
        for h in self._h:
            check(_lib.lib().lora_slab_sync(h), "lora_slab_sync")

    def store(self, out: np.ndarray) -> np.ndarray:
        assert out.flags.c_contiguous
        for h in self._h:
            check(_lib.lib().lora_slab_store(h, out.ctypes.data), "lora_slab_store")
        return out


def run_host_multi(shape, in_: np.ndarray, ngpus: int, params=None, times: int = 1, quiet: bool = True):
    """lora_run_host_multi: the host-buffer operator on `ngpus` devices (or slabs, under LORA_SLAB_LOOPBACK=1)."""
    sid = ops.shape_id(shape)
    bf16 = in_.dtype == np.uint16
    in_ = np.ascontiguousarray(in_ if bf16 else in_.astype(np.float64, copy=False))
    h = ops.halo(sid)
    dims = [in_.shape[i] - 2 * h[i] for i in range(in_.ndim)]
    out = np.zeros_like(in_)
    pp = None if params is None else np.ascontiguousarray(params, dtype=np.float64).ctypes.data_as(
        ctypes.POINTER(ctypes.c_double))
    info = _lib.RunInfo()
    check(_lib.lib().lora_run_host_multi(sid, _lib.BF16 if bf16 else _lib.F64, in_.ctypes.data, out.ctypes.data, pp,
                                         int(times), ops._dims_arg(dims), int(ngpus), int(quiet), ctypes.byref(info)),
          "lora_run_host_multi")
    return out, info
