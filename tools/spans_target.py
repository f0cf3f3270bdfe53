#!/usr/bin/env python3
"""A few launches of the register-resident 3D kernel with option spans3 = argv[1] (for tools/profile_kernel.sh):
   python tools/spans_target.py <0|1> [dtype] [ZxYxX] [shape]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402

spans = int(sys.argv[1])
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dims = tuple(int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "768x768x768").split("x"))
shape = sys.argv[4] if len(sys.argv) > 4 else "box3d1r"
tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
src = (torch.rand(L.padded_shape(shape, dims), device="cuda") * 2 - 1).to(tdt)
dst = src.clone()
plan = L.Plan(shape, dims, dtype=dtype)
plan.set_option("steps_per_launch", 4)
plan.set_option("spans3", spans)
for _ in range(6):
    plan.stepk(src, dst)
torch.cuda.synchronize()
print(plan.kernel_name, "spans3", spans)
