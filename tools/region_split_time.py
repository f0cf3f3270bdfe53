#!/usr/bin/env python3
"""What the deferred wait's region split costs a thin 3D slab: one launch over all planes against interior + two ends
(three launches), interior + one launch of the ends' size, and interior + both ends in ONE launch (lora_plan_stepn_region2).
   python tools/region_split_time.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for shape, dims, dtype in (("star3d1r", (72, 512, 512), "f64"), ("box3d1r", (104, 768, 768), "f64"), ("box3d1r", (104, 768, 768), "bf16"),
                           ("star3d1r", (136, 512, 512), "f64")):
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
    src = (torch.rand(L.padded_shape(shape, dims), device="cuda") * 2 - 1).to(tdt)
    dst = src.clone()
    plan = L.Plan(shape, dims, dtype=dtype)
    plan.set_option("steps_per_launch", 4)
    h, need = dims[0], 4
    a, b = 2 * need, h - 2 * need  # own planes [need, h - need); deep interior [2 need, h - 2 need)
    row = {"shape": shape, "dims": dims, "dtype": dtype, "kernel": plan.kernel_name}
    row["whole_us"] = round(timed(lambda: plan.stepk_region(src, dst, need, h - need)), 1)
    row["three_regions_us"] = round(timed(lambda: (plan.stepk_region(src, dst, a, b), plan.stepk_region(src, dst, need, a),
                                                     plan.stepk_region(src, dst, b, h - need))), 1)
    row["interior_plus_one_end_launch_us"] = round(timed(lambda: (plan.stepk_region(src, dst, a, b), plan.stepk_region(src, dst, 0, 2 * need))), 1)
    row["interior_plus_both_ends_in_one_launch_us"] = round(timed(lambda: (plan.stepk_region(src, dst, a, b),
                                                                           plan.stepn_region2(4, src, dst, need, a, b, h - need))), 1)
    row["interior_us"] = round(timed(lambda: plan.stepk_region(src, dst, a, b)), 1)
    row["one_end_us"] = round(timed(lambda: plan.stepk_region(src, dst, need, a)), 1)
    print(json.dumps(row), flush=True)
