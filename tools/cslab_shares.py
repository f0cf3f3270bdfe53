#!/usr/bin/env python3
"""One rank's share of an N-GPU run through the C++ slab driver (csrc/slab.cpp) as a RING OF ONE over RCCL on one GPU --
tools/slab_shares.py for the driver that has no Python between launches.  star2d1r 16384^2 (rows / N), star3d1r 512^3 and
box3d1r 768^3 fp64 (planes / N); exchange interval E swept.  Projection = N x the per-rank rate (every cost but the link).
Writes gpurun_out/cslab_shares.jsonl."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lorastencil_amd as L
from lorastencil_amd import cslab, _lib

rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
comm = ctypes.c_void_p()
dev = (ctypes.c_int * 1)(0)
assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, dev) == 0
out = open(os.path.join(ROOT, "gpurun_out", "cslab_shares.jsonl"), "a")

def best(fn, n=3):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b

cases = []
for ngpu in (8, 4, 2, 1):
    for e in ((0, 1, 2, 4, 8) if ngpu > 1 else (0,)):
        cases.append(("star2d1r", (16384, 16384), ngpu, e, 96))
for ngpu in (8, 4):
    for e in (0, 1, 2, 4):
        cases.append(("star3d1r", (512, 512, 512), ngpu, e, 48))
        cases.append(("box3d1r", (768, 768, 768), ngpu, e, 48))
for shape, gdims, ngpu, e, steps in cases:
    dims = (gdims[0] // ngpu,) + tuple(gdims[1:])
    w = L.effective_weights(shape)
    w = w / w.sum()
    ring = ngpu > 1
    comms = (_lib.SlabComm * 1)(cslab.rccl_comm(comm.value)) if ring else None
    s = cslab.SlabSet(shape, dims, 1, comms=comms, exchange_every=e, weights=w, flags=cslab.SLAB_RING_OF_ONE if ring else 0)
    a = np.random.default_rng(1).random(L.padded_shape(shape, dims))
    s.load(a)
    def run():
        s.run(steps); s.sync()
    run()
    t = best(run)
    pts = 1
    for d in dims:
        pts *= d
    si = s.info(0)
    rec = {"driver": "csrc/slab.cpp", "shape": shape, "gpus": ngpu, "share": dims, "exchange_every": si.exchange_every, "ghost": si.ghost,
           "apps_per_launch": si.apps_per_launch, "exchanges": si.exchanges, "gstencils_per_rank": round(pts * steps / t / 1e9, 1),
           "projected_gstencils": round(pts * steps / t / 1e9 * ngpu, 1)}
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n"); out.flush()
    s.close()
rccl.ncclCommDestroy(comm)
