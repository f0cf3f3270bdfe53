#!/usr/bin/env python3
"""3D z-slabs of an 8- / 4-GPU run through the C++ driver as a ring of one over RCCL: four sweeps per launch (the
register-resident kernel) against two, strips first against whole slabs.  Writes gpurun_out/cslab_3d_k.jsonl."""
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
out = open(os.path.join(ROOT, "gpurun_out", "cslab_3d_k.jsonl"), "a")

def best(fn, n=3):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b

for shape, dims in (("star3d1r", (64, 512, 512)), ("star3d1r", (128, 512, 512)), ("box3d1r", (96, 768, 768)), ("box3d1r", (192, 768, 768))):
    w = L.effective_weights(shape); w = w / w.sum()
    a = np.random.default_rng(1).random(L.padded_shape(shape, dims))
    pts = int(np.prod(dims))
    for k in (4, 2):
        for e in (1, 2, 4):
            for fl in (0, cslab.SLAB_NO_OVERLAP, cslab.SLAB_NO_OVERLAP | cslab.SLAB_NO_DEFER):
                comms = (_lib.SlabComm * 1)(cslab.rccl_comm(comm.value))
                s = cslab.SlabSet(shape, dims, 1, comms=comms, exchange_every=e, weights=w, flags=cslab.SLAB_RING_OF_ONE | fl,
                                  options={"steps_per_launch": k})
                s.load(a)
                def run():
                    s.run(48); s.sync()
                run()
                t = best(run)
                si = s.info(0)
                rec = {"shape": shape, "share": dims, "apps": si.apps_per_launch, "exchange_every": si.exchange_every, "ghost": si.ghost,
                       "no_overlap": bool(fl & cslab.SLAB_NO_OVERLAP), "no_defer": bool(fl & cslab.SLAB_NO_DEFER),
                       "gstencils_per_rank": round(pts * 48 / t / 1e9, 1)}
                print(json.dumps(rec), flush=True)
                out.write(json.dumps(rec) + "\n"); out.flush()
                s.close()
rccl.ncclCommDestroy(comm)
