#!/usr/bin/env python3
"""The periodic option: fused launches on a ghost-extended grid (run_torus) against single sweeps behind a wrap each.
   python tools/torus_time.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402

for shape, dims, dtype, times in (("star2d1r", (16384, 16384), "f64", 24), ("box2d3r", (8192, 8192), "f64", 24),
                                  ("star3d1r", (512, 512, 512), "f64", 24), ("box3d1r", (768, 768, 768), "f64", 24),
                                  ("box3d1r", (768, 768, 768), "bf16", 24), ("1d1r", (1 << 20,), "f64", 96)):
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
    b0 = (torch.rand(L.padded_shape(shape, dims), device="cuda") * 2 - 1).to(tdt)
    b1 = torch.zeros_like(b0)
    row = {"shape": shape, "dims": dims, "dtype": dtype, "sweeps": times}
    for name, torus in (("torus", 1), ("single_sweeps", 0)):
        plan = L.Plan(shape, dims, dtype=dtype).set_boundary("periodic").set_option("torus", torus)
        plan.run(b0, b1, times)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.run(b0, b1, times)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        pts = 1
        for d in dims:
            pts *= d
        row[name + "_ms"] = round(ms, 3)
        row[name + "_gst"] = round(pts * times / ms / 1e6, 1)
    print(json.dumps(row), flush=True)
    del b0, b1
    torch.cuda.empty_cache()
