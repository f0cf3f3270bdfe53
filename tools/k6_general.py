import sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
import lorastencil_amd as L
rng = np.random.default_rng(3)
for dims in ((8192, 8192), (16384, 16384)):
    w = rng.random(49); w /= w.sum()
    src = (torch.rand(L.padded_shape("box2d3r", dims), device="cuda", dtype=torch.float64) * 2 - 1)
    dst = src.clone()
    for opts in ({}, {"steps_per_launch": 4}, {"steps_per_launch": 6}, {"steps_per_launch": 2}):
        plan = L.Plan("box2d3r", dims).set_weights(w)
        for k, v in opts.items(): plan.set_option(k, v)
        apps = plan.get_option("steps_per_launch")
        for _ in range(3): plan.stepk(src, dst)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan.stepk(src, dst)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        print(json.dumps({"dims": dims, "opts": opts, "apps": apps, "kernel": plan.kernel_name, "sig": plan.kernel_signature if hasattr(plan, "kernel_signature") else None, "us": round(us, 1), "gst": round(dims[0] * dims[1] * apps / us / 1e3, 1)}), flush=True)
