#!/usr/bin/env python3
"""Spans against chunks (option spans3) of the register-resident 3D kernels over region depths, on the GPU box:
   python tools/spans_sweep.py [--dtype bf16|f64] [--shape box3d1r]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402


def time_plan(plan, src, dst, n=12):
    for _ in range(3):
        plan.stepk(src, dst)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan.stepk(src, dst)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--shape", default="box3d1r")
    ap.add_argument("--grids", default="32x768x768,64x768x768,96x768x768,192x768x768,384x768x768,768x768x768,64x512x512,128x512x512,512x512x512,256x1024x1024")
    args = ap.parse_args()
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float64
    for g in args.grids.split(","):
        dims = tuple(int(x) for x in g.split("x"))
        ps = L.padded_shape(args.shape, dims)
        gen = torch.Generator(device="cuda").manual_seed(7)
        src = (torch.rand(ps, generator=gen, device="cuda") * 2 - 1).to(tdt)
        dst = src.clone()
        pts = dims[0] * dims[1] * dims[2]
        row = {"shape": args.shape, "dtype": args.dtype, "dims": dims}
        for name, v in (("warm", 0), ("chunks", 0), ("spans", 1), ("teams", 2), ("rule", -1), ("chunks_again", 0), ("spans_again", 1),
                        ("teams_again", 2), ("rule_again", -1)):  # (the first block runs in the clock ramp: not reported)
            plan = L.Plan(args.shape, dims, dtype=args.dtype)
            plan.set_option("steps_per_launch", 4)
            plan.set_option("spans3", v)
            us = time_plan(plan, src, dst)
            if name == "warm":
                continue
            row[name + "_us"] = round(us, 1)
            row["kernel"] = plan.kernel_name
        outs = []
        for v in (0, 1, 2):  # the cuts of the launch give the same bits
            plan = L.Plan(args.shape, dims, dtype=args.dtype)
            plan.set_option("steps_per_launch", 4)
            plan.set_option("spans3", v)
            o = src.clone()
            plan.stepk(src, o)
            torch.cuda.synchronize()
            outs.append(o)
        it = torch.int16 if args.dtype == "bf16" else torch.int64
        row["same_bits"] = bool(torch.equal(outs[0].view(it), outs[1].view(it)) and torch.equal(outs[0].view(it), outs[2].view(it)))
        del outs
        print(json.dumps(row), flush=True)
        del src, dst
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
