#!/usr/bin/env python3
"""bf16 z-slabs (BASELINE config 5: box3d1r 768^3 bf16) through the C++ driver as a ring of one over RCCL: whole slab per
launch (the default) against boundary planes first.  Writes gpurun_out/cslab_bf16.jsonl."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lorastencil_amd as L
from lorastencil_amd import cslab, _lib
from oracle import oracle as O  # (bf16 conversion of the input only)

rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
comm = ctypes.c_void_p()
dev = (ctypes.c_int * 1)(0)
assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, dev) == 0
out = open(os.path.join(ROOT, "gpurun_out", "cslab_bf16.jsonl"), "a")

def best(fn, n=3):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b

shape = "box3d1r"
w = L.effective_weights(shape); w = w / w.sum()
for ngpu in (8, 4, 2, 1):
    dims = (768 // ngpu, 768, 768)
    bits = O.to_bf16(np.random.default_rng(1).random(L.padded_shape(shape, dims)))
    for e in ((0, 1, 2, 4) if ngpu > 1 else (0,)):
        for fl in ((0, cslab.SLAB_OVERLAP, cslab.SLAB_NO_DEFER) if ngpu > 1 else (0,)):
            ring = ngpu > 1
            comms = (_lib.SlabComm * 1)(cslab.rccl_comm(comm.value)) if ring else None
            s = cslab.SlabSet(shape, dims, 1, comms=comms, exchange_every=e, weights=w, dtype="bf16", flags=(cslab.SLAB_RING_OF_ONE if ring else 0) | fl)
            s.load(bits)
            def run():
                s.run(48); s.sync()
            run()
            t = best(run)
            si = s.info(0)
            rec = {"shape": shape, "dtype": "bf16", "gpus": ngpu, "share": dims, "exchange_every": si.exchange_every, "ghost": si.ghost,
                   "apps": si.apps_per_launch, "strips_first": bool(fl & cslab.SLAB_OVERLAP), "no_defer": bool(fl & cslab.SLAB_NO_DEFER),
                   "gstencils_per_rank": round(int(np.prod(dims)) * 48 / t / 1e9, 1)}
            rec["projected_gstencils"] = round(rec["gstencils_per_rank"] * ngpu, 1)
            print(json.dumps(rec), flush=True)
            out.write(json.dumps(rec) + "\n"); out.flush()
            s.close()
rccl.ncclCommDestroy(comm)
