#!/usr/bin/env python3
"""Microseconds per launch of the 3D register-resident kernels at four and at two applications (the tail of a run):
   python tools/tail_time.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402


def timed(fn, n=12):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for shape, dims, dtype in (("star3d1r", (512, 512, 512), "f64"), ("box3d1r", (768, 768, 768), "f64"), ("box3d1r", (768, 768, 768), "bf16")):
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
    src = (torch.rand(L.padded_shape(shape, dims), device="cuda") * 2 - 1).to(tdt)
    dst = src.clone()
    plan = L.Plan(shape, dims, dtype=dtype)
    row = {"shape": shape, "dims": dims, "dtype": dtype, "kernel": plan.kernel_name, "apps": plan.get_option("steps_per_launch")}
    row["warm_us"] = round(timed(lambda: plan.stepk(src, dst)), 1)
    row["four_us"] = round(timed(lambda: plan.stepk(src, dst)), 1)
    row["two_us"] = round(timed(lambda: plan.step2(src, dst)), 1)
    row["one_us"] = round(timed(lambda: plan.step(src, dst)), 1)
    p2 = L.Plan(shape, dims, dtype=dtype).set_option("lanes3", 0)
    row["two_us_tile_kernel"] = round(timed(lambda: p2.step2(src, dst)), 1)
    row["tile_kernel"] = p2.kernel_name
    print(json.dumps(row), flush=True)
