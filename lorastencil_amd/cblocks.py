"""ctypes mirror of the C++ block driver (include/lorastencil.h, group E; csrc/blocks.cpp): a Pa x Pb decomposition of the
two outer dimensions -- rows x columns in 2D, planes x rows in 3D -- with ghost zones on every cut side.  `BlockGrid` holds
every block of a decomposition in ONE process (loopback exchange on one device, or RCCL communicators from
ncclCommInitAll on several); one block per process is `Block` with an RCCL table."""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np

from . import _lib, ops
from .cslab import loopback_comms  # noqa: F401  (the same callback tables as the slabs)


class Block:
    """One block of a Pa x Pb grid (lora_block)."""

    def __init__(self, shape, global_dims: Sequence[int], grid: Sequence[int], coords: Sequence[int], comm=None, device: int = 0,
                 dtype="f64", params=None, weights=None, exchange_every: int = 0, flags: int = 0, options=None):
        self.shape = ops.shape_id(shape)
        self.dtype = ops.dtype_id(dtype)
        d = _lib.BlockDesc()
        d.shape, d.dtype, d.device, d.exchange_every, d.flags = self.shape, self.dtype, int(device), int(exchange_every), int(flags)
        for k, v in enumerate(global_dims):
            d.global_dims[k] = int(v)
        d.grid[0], d.grid[1] = int(grid[0]), int(grid[1])
        d.coords[0], d.coords[1] = int(coords[0]), int(coords[1])
        self._keep = []
        if params is not None:
            p = np.ascontiguousarray(params, dtype=np.float64)
            d.params = p.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            self._keep.append(p)
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float64)
            d.weights = w.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            self._keep.append(w)
        if options:
            o = ",".join(f"{k}={int(v)}" for k, v in options.items()).encode()
            d.options = o
            self._keep.append(o)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().lora_block_create(ctypes.byref(self._h), ctypes.byref(d), ctypes.byref(comm) if comm is not None else None),
                   "lora_block_create")

    def info(self):
        i = _lib.BlockInfo()
        _lib.check(_lib.lib().lora_block_info(self._h, ctypes.byref(i)), "lora_block_info")
        return i

    def load(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a)
        _lib.check(_lib.lib().lora_block_load(self._h, a.ctypes.data_as(ctypes.c_void_p)), "lora_block_load")

    def run(self, times: int) -> None:
        _lib.check(_lib.lib().lora_block_run(self._h, int(times)), "lora_block_run")

    def sync(self) -> None:
        _lib.check(_lib.lib().lora_block_sync(self._h), "lora_block_sync")

    def store(self, out: np.ndarray) -> np.ndarray:
        assert out.flags["C_CONTIGUOUS"]
        _lib.check(_lib.lib().lora_block_store(self._h, out.ctypes.data_as(ctypes.c_void_p)), "lora_block_store")
        return out

    def close(self) -> None:
        if self._h:
            _lib.lib().lora_block_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BlockGrid:
    """Every block of a decomposition, driven from this process (lora_block_run_many)."""

    def __init__(self, shape, global_dims, grid, comms=None, devices=None, **kw):
        pa, pb = int(grid[0]), int(grid[1])
        n = pa * pb
        self.blocks = [Block(shape, global_dims, grid, (r // pb, r % pb), comm=(comms[r] if comms is not None else None),
                             device=(devices[r] if devices is not None else 0), **kw) for r in range(n)]
        self._arr = (ctypes.c_void_p * n)(*[b._h for b in self.blocks])

    def info(self, r: int = 0):
        return self.blocks[r].info()

    def load(self, a) -> None:
        for b in self.blocks:
            b.load(a)

    def run(self, times: int) -> None:
        _lib.check(_lib.lib().lora_block_run_many(self._arr, len(self.blocks), int(times)), "lora_block_run_many")

    def sync(self) -> None:
        for b in self.blocks:
            b.sync()

    def store(self, out: np.ndarray) -> np.ndarray:
        for b in self.blocks:
            b.store(out)
        return out

    def close(self) -> None:
        for b in self.blocks:
            b.close()


def run_host_blocks(shape, a: np.ndarray, grid: Sequence[int], times: int, dtype="f64", params=None):
    """lora_run_host_blocks: the host-buffer operator on a grid[0] x grid[1] grid of blocks (one per device; all on device 0
    under LORA_SLAB_LOOPBACK=1).  Returns (out, lora_run_info)."""
    sid = ops.shape_id(shape)
    a = np.ascontiguousarray(a)
    out = np.zeros_like(a)
    nd = ops.ndim(shape)
    halo = ops.halo(shape)
    dims = (ctypes.c_int * 3)(*[a.shape[k] - 2 * halo[k] for k in range(nd)] + [0] * (3 - nd))
    g = (ctypes.c_int * 2)(int(grid[0]), int(grid[1]))
    info = _lib.RunInfo()
    pp = None
    if params is not None:
        p = np.ascontiguousarray(params, dtype=np.float64)
        pp = p.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    _lib.check(_lib.lib().lora_run_host_blocks(sid, ops.dtype_id(dtype), a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), pp,
                                              int(times), dims, g, 1, ctypes.byref(info)), "lora_run_host_blocks")
    return out, info
