#!/usr/bin/env python3
"""Development tool: A/B of two builds of the engine in one run (python tools/wg_ab.py [path/to/other/liblorastencil_hip.so]):
the six-sweep 2D launch on full and thin grids.  The library path is patched before the first load -- a tool-only hook."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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

for shape, dims in (("star2d1r", (16384, 16384)), ("star2d1r", (2048, 16384)), ("star2d1r", (4096, 16384)), ("box2d3r", (8192, 8192)),
                    ("star2d1r", (4096, 4096)), ("star2d3r", (16384, 16384))):
    w = L.effective_weights(shape); w = w / w.sum()
    ps = L.padded_shape(shape, dims)
    src = torch.rand(ps, dtype=torch.float64, device="cuda"); dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
    plan = L.Plan(shape, dims).set_weights(w)
    K = plan.get_option("steps_per_launch")
    t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), 20) / 2
    print(json.dumps({"lib": os.path.basename(os.path.dirname(_lib.LIB_PATH)), "shape": shape, "dims": dims, "K": K, "kernel": plan.kernel_name,
                      "us": round(t * 1e6, 1), "gstencils": round(dims[0] * dims[1] * K / t / 1e9, 1)}), flush=True)
