#!/usr/bin/env python3
"""Kernel-option sweep on one GPU (development tool): times every (shape, option) combination with HIP events in
one process and prints a table plus JSON lines (gpurun_out/sweep.jsonl).  Also measures the device copy bandwidth
(torch copy_) of the same byte count as the reference point for "achievable" HBM bandwidth."""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402


def time_fn(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.jsonl"))
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    dev = torch.device("cuda", 0)
    rows = []

    def record(**kw):
        rows.append(kw)
        print(json.dumps(kw), flush=True)
        with open(args.out, "a") as f:
            f.write(json.dumps(kw) + "\n")

    # ---- achievable bandwidth: device-to-device copy of one 16384^2 padded grid
    a = torch.empty((16392, 16392), dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    t = time_fn(lambda: b.copy_(a), 20)
    record(kind="copy", bytes=2 * a.numel() * 8, seconds=t, gbs=2 * a.numel() * 8 / t / 1e9)
    del a, b

    cases = [
        ("star2d1r", (16384, 16384), {"steps_per_launch": [2], "lowrank_valu": [0, 1], "fused_rows": [6, 8, 10]}),
        ("box2d3r", (8192, 8192), {"steps_per_launch": [2], "lowrank_valu": [0, 1], "fused_rows": [6, 8, 10]}),
        ("star2d1r", (16384, 16384), {"steps_per_launch": [1], "rows_per_thread": [4, 8], "panel_width": [16, 32], "variant": [1]}),
        ("star2d1r", (16384, 16384), {"panel_width": [16, 32], "variant": [2]}),
        ("box2d3r", (8192, 8192), {"steps_per_launch": [1], "rows_per_thread": [4, 8], "panel_width": [8, 16, 32, 64], "variant": [1]}),
        ("box2d3r", (8192, 8192), {"panel_width": [16, 32], "variant": [2]}),
        ("star2d3r", (16384, 16384), {"steps_per_launch": [1], "rows_per_thread": [4, 8], "panel_width": [16, 32], "variant": [1]}),
        ("star2d3r", (16384, 16384), {"steps_per_launch": [2], "fused_rows": [6, 8, 10]}),
        ("star2d3r", (16384, 16384), {"panel_width": [32], "variant": [2]}),
        ("star3d1r", (512, 512, 512), {"steps_per_launch": [1], "z_chunk": [4, 7, 16, 31, 64]}),
        ("box3d1r", (768, 768, 768), {"steps_per_launch": [1], "z_chunk": [7, 16, 31]}),
        ("star3d1r", (64, 512, 512), {"steps_per_launch": [1], "z_chunk": [4, 7, 16]}),
        # 3D temporal fusion: output planes per workgroup (every chunk re-reads 4 planes)
        ("star3d1r", (512, 512, 512), {"steps_per_launch": [2], "fused_z_chunk": [8, 12, 16, 24, 32, 64]}),
        ("box3d1r", (768, 768, 768), {"steps_per_launch": [2], "fused_z_chunk": [8, 16, 24, 32, 64]}),
        ("star3d1r", (64, 512, 512), {"steps_per_launch": [2], "fused_z_chunk": [0, 8, 16, 32]}),
        ("box3d1r", (768, 768, 768), {"steps_per_launch": [1], "separable": [0, 1]}, "bf16"),
        ("box3d1r", (768, 768, 768), {"steps_per_launch": [2], "separable": [0, 1],
                                      "fused_z_chunk": [12, 16, 24, 32, 48, 64]}, "bf16"),
        ("box3d1r", (768, 768, 768), {"steps_per_launch": [2], "fused_pipeline": [0, 1]}, "bf16"),
        ("star3d1r", (768, 768, 768), {"steps_per_launch": [1, 2]}, "bf16"),
        ("1d1r", (1048576,), {"steps_per_launch": [1, 2, 4, 8]}),
        ("1d1r", (1 << 28,), {"steps_per_launch": [1, 2, 4, 8]}),
    ]
    for case in cases:
        shape, dims, grid = case[:3]
        dtype = case[3] if len(case) > 3 else "f64"
        if args.only and args.only not in shape and args.only != dtype:
            continue
        if args.only == "3d" and len(dims) != 3:
            continue
        plan = L.Plan(shape, dims, dtype=dtype)
        esize = 2 if dtype == "bf16" else 8
        w = plan.weights
        plan.set_weights(w / w.sum())  # normalised taps: values stay bounded however many sweeps are timed
        ps = plan.padded_shape
        src = torch.randint(0, 100, ps, device=dev).to(torch.bfloat16 if dtype == "bf16" else torch.float64)
        dst = torch.zeros_like(src)
        pts = 1
        for d in dims:
            pts *= d
        keys = list(grid)
        combos = list(itertools.product(*[grid[k] for k in keys])) or [()]
        for combo in combos:
            for k, v in zip(keys, combo):
                if k == "variant":
                    plan.set_variant(v)
                else:
                    plan.set_option(k, v)
            iters = 5 if args.quick else 20
            spl = plan.get_option("steps_per_launch")
            if plan.get_option("variant") == 2:
                spl = 1  # the MFMA variant is a single-sweep kernel
            if spl >= 2:
                t = time_fn(lambda: plan.stepk(src, dst), iters) / spl  # per application
            else:
                t = time_fn(lambda: plan.step(src, dst), iters)
            record(kind="sweep", shape=shape, dims=dims, dtype=dtype, options=dict(zip(keys, combo)),
                   kernel=plan.kernel_name, seconds=t, gstencils=pts / t / 1e9, algo_gbs=pts * 2 * esize / t / 1e9,
                   frac_of_8TBs=pts * 2 * esize / t / 8e12)
        del src, dst
        torch.cuda.empty_cache()

    best = {}
    for r in rows:
        if r["kind"] == "sweep":
            key = (r["shape"], tuple(r["dims"]), r.get("dtype", "f64"))
            if key not in best or r["gstencils"] > best[key]["gstencils"]:
                best[key] = r
    print("\nBEST per case:")
    for k, r in best.items():
        print(f"  {k[0]:10s} {str(k[1]):22s} {k[2]:5s} {r['options']}  {r['gstencils']:.1f} GSt/s  {r['algo_gbs']:.0f} GB/s  "
              f"{100 * r['frac_of_8TBs']:.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
