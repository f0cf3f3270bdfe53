#!/usr/bin/env python3
"""Does the RCCL exchange of a slab overlap the interior launch?  The six-sweep 2D kernel fills every CU with its one
round of workgroups; RCCL's send / recv kernels then wait for a free slot.  Ring of one over RCCL, 2048 x 16384 share of
star2d1r, E = 1 / 2: default chunking against longer chunks (wg_rows: fewer workgroups than resident slots, some CUs keep
room).  Writes gpurun_out/cslab_overlap.jsonl."""
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
out = open(os.path.join(ROOT, "gpurun_out", "cslab_overlap.jsonl"), "a")

def best(fn, n=3):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b

for shape, dims, steps in (("star2d1r", (2048, 16384), 96), ("star2d1r", (4096, 16384), 96), ("star2d1r", (8192, 16384), 96),
                           ("star3d1r", (64, 512, 512), 48), ("star3d1r", (128, 512, 512), 48), ("box3d1r", (96, 768, 768), 48),
                           ("box3d1r", (192, 768, 768), 48)):
    w = L.effective_weights(shape); w = w / w.sum()
    a = np.random.default_rng(1).random(L.padded_shape(shape, dims))
    pts = int(np.prod(dims))
    for backend in ("rccl",):
        for e in ((2, 4, 8) if len(dims) == 2 else (1, 2, 4)):
            for fl in (0, cslab.SLAB_NO_OVERLAP, cslab.SLAB_NO_DEFER, cslab.SLAB_NO_OVERLAP | cslab.SLAB_NO_DEFER):
                comms = (_lib.SlabComm * 1)(cslab.rccl_comm(comm.value))
                s = cslab.SlabSet(shape, dims, 1, comms=comms, exchange_every=e, weights=w, flags=cslab.SLAB_RING_OF_ONE | fl)
                s.load(a)
                def run():
                    s.run(steps); s.sync()
                run()
                t = best(run)
                si = s.info(0)
                rec = {"shape": shape, "share": dims, "exchange_every": si.exchange_every, "ghost": si.ghost, "apps": si.apps_per_launch,
                       "no_overlap": bool(fl & cslab.SLAB_NO_OVERLAP), "no_defer": bool(fl & cslab.SLAB_NO_DEFER),
                       "gstencils_per_rank": round(pts * steps / t / 1e9, 1)}
                print(json.dumps(rec), flush=True)
                out.write(json.dumps(rec) + "\n"); out.flush()
                s.close()
rccl.ncclCommDestroy(comm)
