#!/usr/bin/env python3
"""Development tool: chunk length (wg_rows) of the six-sweep 2D kernel around its one-round default on the BASELINE grids."""
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

for shape, dims, rows_list in (("star2d1r", (16384, 16384), (0, 1000, 1100, 1177, 1200, 1240, 1280, 1320, 1380, 1450, 1550, 1700, 2048)),
                               ("star2d1r", (8192, 8192), (0, 260, 292, 300, 310, 320, 330, 345, 360, 380, 410, 450)),
                               ("box2d3r", (8192, 8192), (0, 260, 292, 300, 310, 320, 330, 345, 360, 380, 410, 450))):
    w = L.effective_weights(shape); w = w / w.sum()
    ps = L.padded_shape(shape, dims)
    src = torch.rand(ps, dtype=torch.float64, device="cuda"); dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
    for rows in rows_list:
        plan = L.Plan(shape, dims).set_weights(w)
        if rows:
            plan.set_option("wg_rows", rows)
        K = plan.get_option("steps_per_launch")
        t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), 20) / 2
        print(json.dumps({"shape": shape, "dims": dims, "wg_rows": rows, "us": round(t * 1e6, 1), "gstencils": round(dims[0] * dims[1] * K / t / 1e9, 1)}), flush=True)
