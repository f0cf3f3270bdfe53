#!/usr/bin/env python3
"""Would z x y blocks beat z-slabs for the 3D configurations at 8 GPUs?  The device side of the question, on one GPU: the
fused launch of one rank's LOCAL grid (own cells + ghost zones of 4 x E cells on every cut side) for the slab and the block
layouts of star3d1r 512^3 / box3d1r 768^3 (fp64) and box3d1r 768^3 bf16; GStencils/s counted on OWN cells only.
    python tools/block3d_shares.py > gpurun_out/block3d_shares.jsonl"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402

E = 2  # launches between refreshes: ghost = radius x 4 applications x E = 8 cells per cut side


def rate(shape, local, own_pts, dtype):
    w = [x / sum(L.effective_weights(shape)) for x in L.effective_weights(shape)]
    plan = L.Plan(shape, local, dtype=dtype).set_weights(w).set_option("steps_per_launch", 4)
    ps = L.padded_shape(shape, local)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
    src = (torch.rand(ps, device="cuda") * 2 - 1).to(tdt)
    dst = src.clone()
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
    apps = plan.get_option("steps_per_launch")
    return {"kernel": plan.kernel_name, "apps": apps, "us": round(us, 1), "gstencils_own": round(own_pts * apps / us / 1e3, 1)}


for shape, g, dtype in (("star3d1r", 512, "f64"), ("box3d1r", 768, "f64"), ("box3d1r", 768, "bf16")):
    gh = 4 * E
    for name, (pz, py) in (("8 x 1 slabs", (8, 1)), ("4 x 2 blocks", (4, 2)), ("2 x 4 blocks", (2, 4))):
        own = (g // pz, g // py, g)
        # a middle rank: ghost zones on both sides of every cut dimension
        local = (own[0] + (2 * gh if pz > 1 else 0), own[1] + (2 * gh if py > 1 else 0), g)
        own_pts = own[0] * own[1] * own[2]
        rec = {"shape": shape, "dtype": dtype, "global": g, "layout": name, "own": own, "local": local,
               "redundant_cells_pct": round(100.0 * (local[0] * local[1] * local[2] / own_pts - 1), 1),
               "ghost_bytes_per_refresh": (2 * gh * own[1] * g * (pz > 1) + 2 * gh * local[0] * g * (py > 1)) * (2 if dtype == "bf16" else 8)}
        rec.update(rate(shape, local, own_pts, dtype))
        print(json.dumps(rec), flush=True)
