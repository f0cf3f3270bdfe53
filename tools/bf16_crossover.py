#!/usr/bin/env python3
"""bf16 box3d1r: GStencils/s per launch of the register-resident kernel (four sweeps) and the tile kernel (two) over grid
sizes -- where the plan should switch (capi.cpp plan_refresh).   python tools/bf16_crossover.py > gpurun_out/..."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402

shape = "box3d1r"
w = [x / 36.0 for x in L.effective_weights(shape)]
for dims in ((128, 128, 128), (192, 192, 192), (256, 256, 256), (320, 320, 320), (384, 384, 384), (512, 512, 512), (48, 768, 768), (64, 768, 768),
             (96, 768, 768), (192, 768, 768), (384, 768, 768), (768, 768, 768), (256, 512, 1024)):
    ps = L.padded_shape(shape, dims)
    src = (torch.rand(ps, device="cuda") * 2 - 1).to(torch.bfloat16)
    dst = src.clone()
    pts = dims[0] * dims[1] * dims[2]
    rec = {"dims": dims, "points": pts}
    for name, opts in (("lanes4", {"steps_per_launch": 4}), ("fused2", {"lanes3": 0})):
        plan = L.Plan(shape, dims, dtype="bf16").set_weights(w)
        for k, v in opts.items():
            plan.set_option(k, v)
        apps = plan.get_option("steps_per_launch")
        for _ in range(3):
            plan.stepk(src, dst)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            plan.stepk(src, dst)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        rec[name] = {"kernel": plan.kernel_name, "apps": apps, "us": round(us, 1), "gstencils": round(pts * apps / us / 1e3, 1)}
    print(json.dumps(rec), flush=True)
