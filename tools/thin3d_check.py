#!/usr/bin/env python3
"""Development tool: the 3D fp64 kernels on THIN grids (what a z-slab of an 8-GPU run is): per-launch rates of every fused
kernel on h x 512 x 512 / h x 768 x 768 for small h, no slab driver around them."""
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

for shape, base in (("star3d1r", 512), ("box3d1r", 768)):
    for h in (64, 72, 80, 96, 112, 128, 144, 192, 256, 384, base):
        dims = (h, base, base)
        w = L.effective_weights(shape); w = w / w.sum()
        ps = L.padded_shape(shape, dims)
        src = torch.rand(ps, dtype=torch.float64, device="cuda"); dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
        full = len(sys.argv) > 1 and sys.argv[1] == "--all"
        for opts in (({"steps_per_launch": 4}, {"steps_per_launch": 4, "fused_z_chunk": h}, {"steps_per_launch": 4, "fused_z_chunk": (h + 1) // 2}, {"steps_per_launch": 2, "lanes3": 1}, {"steps_per_launch": 2, "lanes3": 0}, {"steps_per_launch": 3, "lanes3": 0}) if full else ({"steps_per_launch": 4}, {"steps_per_launch": 2, "lanes3": 0})):
            try:
                plan = L.Plan(shape, dims).set_weights(w)
                for k, v in opts.items():
                    plan.set_option(k, v)
                K = plan.get_option("steps_per_launch")
                t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), 20) / 2
            except Exception as e:  # noqa: BLE001
                print("skip", shape, dims, opts, e, flush=True); continue
            print(json.dumps({"shape": shape, "dims": dims, "opts": opts, "K": K, "kernel": plan.kernel_name, "us": round(t * 1e6, 1),
                              "gstencils": round(h * base * base * K / t / 1e9, 1)}), flush=True)
