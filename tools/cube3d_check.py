#!/usr/bin/env python3
"""Development tool: where does the register-resident 3D kernel (four sweeps per launch) overtake the LDS-tile kernels
(three / two)?  Cubes and slabs of increasing size, per-sweep rates (us per sweep decides, launches fuse differently)."""
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

for shape in ("star3d1r", "box3d1r"):
    for dims in ((96, 96, 96), (128, 128, 128), (160, 160, 160), (192, 192, 192), (224, 224, 224), (256, 256, 256), (320, 320, 320), (384, 384, 384),
                 (32, 512, 512), (48, 512, 512), (64, 256, 256), (128, 256, 256), (512, 128, 128), (256, 512, 128)):
        w = L.effective_weights(shape); w = w / w.sum()
        ps = L.padded_shape(shape, dims)
        src = torch.rand(ps, dtype=torch.float64, device="cuda"); dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
        row = {"shape": shape, "dims": dims, "mpts": round(dims[0] * dims[1] * dims[2] / 1e6, 1)}
        for name, opts in (("lanes4", {"steps_per_launch": 4}), ("tiles", {"lanes3": 0})):
            plan = L.Plan(shape, dims).set_weights(w)
            for k, v in opts.items():
                plan.set_option(k, v)
            K = plan.get_option("steps_per_launch")
            t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), 20) / 2
            row[name] = {"K": K, "kernel": plan.kernel_name.replace("stencil3d_", ""), "gst": round(dims[0] * dims[1] * dims[2] * K / t / 1e9, 1)}
        print(json.dumps(row), flush=True)
