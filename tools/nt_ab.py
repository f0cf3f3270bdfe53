#!/usr/bin/env python3
"""Development tool: two builds of the engine in one run (python tools/nt_ab.py [other liblorastencil_hip.so]): per-launch
times of the fused kernels on the BASELINE grids.  The library path is patched before the first load -- a tool-only hook."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lorastencil_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import lorastencil_amd as L

def time_fn(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters

tag = os.path.basename(os.path.dirname(_lib.LIB_PATH))
for shape, dims, Ks, dtype in (("box3d1r", (768, 768, 768), (2,), "bf16"), ("star3d1r", (512, 512, 512), (4,), "f64"), ("star2d1r", (16384, 16384), (6,), "f64")):
    w = L.effective_weights(shape); w = w / w.sum()
    ps = L.padded_shape(shape, dims)
    if dtype == "bf16":
        src = torch.rand(ps, device="cuda").to(torch.bfloat16); dst = torch.zeros(ps, dtype=torch.bfloat16, device="cuda")
    else:
        src = torch.rand(ps, dtype=torch.float64, device="cuda"); dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
    for K in Ks:
        plan = L.Plan(shape, dims, dtype=dtype).set_weights(w)
        plan.set_option("steps_per_launch", K)
        t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), 20) / 2
        print(json.dumps({"lib": tag, "shape": shape, "dims": dims, "dtype": dtype, "K": K, "kernel": plan.kernel_name, "us": round(t * 1e6, 1),
                          "gstencils": round(int(np.prod(dims)) * K / t / 1e9, 1)}), flush=True)
