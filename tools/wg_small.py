#!/usr/bin/env python3
"""Development tool: chunk length of the six-sweep 2D kernel on small grids (fewer rows than one round of workgroups
wants): wg_rows sweep."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lorastencil_amd as L

def time_fn(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters

for shape, dims in (("star2d1r", (4096, 4096)), ("star2d1r", (2048, 2048)), ("star2d1r", (8192, 8192)), ("box2d3r", (8192, 8192)), ("star2d1r", (2048, 16384)), ("star2d1r", (1024, 16384)), ("star2d1r", (8192, 2048))):
    w = L.effective_weights(shape); w = w / w.sum()
    ps = L.padded_shape(shape, dims)
    src = torch.rand(ps, dtype=torch.float64, device="cuda"); dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
    for rows in (0, 50, 64, 80, 100, 128, 164, 200, 256, 330, 512):
        if rows > dims[0]:
            continue
        plan = L.Plan(shape, dims).set_weights(w)
        if rows:
            plan.set_option("wg_rows", rows)
        K = plan.get_option("steps_per_launch")
        t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), 20) / 2
        print(json.dumps({"shape": shape, "dims": dims, "wg_rows": rows, "K": K, "us": round(t * 1e6, 1), "gstencils": round(dims[0] * dims[1] * K / t / 1e9, 1)}), flush=True)
