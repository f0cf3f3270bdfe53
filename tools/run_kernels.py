#!/usr/bin/env python3
"""Development tool for profiling: launches a list of kernel configurations back to back on the same data so that one
rocprofv3 run attributes its counters per kernel.  Each --case is  name:opt=val,opt=val  (shape/size apply to all)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="star2d1r")
ap.add_argument("--size", type=int, nargs="+", default=[16384, 16384])
ap.add_argument("--dtype", default="f64")
ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--case", action="append", default=[])
args = ap.parse_args()
dims = tuple(args.size)
w = L.effective_weights(args.shape)
w = w / w.sum()
ps = L.padded_shape(args.shape, dims)
tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float64
src = torch.randint(0, 100, ps, device="cuda").to(tdt)
dst = torch.zeros_like(src)
pts = 1
for d in dims:
    pts *= d
for case in args.case or ["default:"]:
    name, _, optstr = case.partition(":")
    plan = L.Plan(args.shape, dims, dtype=args.dtype).set_weights(w)
    for kv in filter(None, optstr.split(",")):
        k, v = kv.split("=")
        plan.set_option(k, int(v))
    apps = plan.get_option("steps_per_launch")
    plan.stepk(src, dst)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.launches // 2):
        plan.stepk(src, dst)
        plan.stepk(dst, src)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 1e3 / (2 * (args.launches // 2))
    print(f'{{"case": "{name}", "kernel": "{plan.kernel_signature}", "us": {t * 1e6:.1f}, "gstencils": {apps * pts / t / 1e9:.1f}}}', flush=True)
