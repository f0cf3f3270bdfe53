#!/usr/bin/env python3
"""Parity and timing of the register-resident bf16 kernel (kernels_3d_bf16_lanes.hip) on the GPU box:
   * runs of 4 .. 13 sweeps with four applications per launch against the bf16 oracle (whole padded buffer) on small,
     ragged and rim-only grids, and against single sweeps of the engine on grids of several tiles and chunks;
   * microseconds per launch at 768^3 beside the two-sweep tile kernel.
     python tools/bf16_lanes_check.py [--quick] [--time-only]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)


def run_bits(shape, bits, times, weights, options):
    dims = (bits.shape[0] - 2, bits.shape[1] - 4, bits.shape[2] - 8)
    plan = L.Plan(shape, dims, dtype="bf16").set_weights(weights)
    for k, v in options.items():
        plan.set_option(k, v)
    b0 = torch.from_numpy(bits.view(np.int16).copy()).cuda().view(torch.bfloat16)
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, times)
    torch.cuda.synchronize()
    return (b0, b1)[times % 2].view(torch.int16).cpu().numpy().view(np.uint16), plan.kernel_name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--time-only", action="store_true")
    args = ap.parse_args()
    shape = "box3d1r"
    w = O.effective_weights(shape)
    w = w / w.sum()
    rng = np.random.default_rng(5)
    bad = 0
    if not args.time_only:
        small = [(3, 5, 8), (8, 16, 128), (9, 31, 248), (40, 61, 128), (37, 64, 360), (21, 120, 136), (12, 57, 121 * 8),
                 (30, 113, 240), (5, 56, 120), (64, 7, 16)]
        for dims in small:
            bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
            for t in (4, 5, 8, 9, 13):
                exp = O.run_bf16(shape, bits, t, weights=w)
                for opts in ({"steps_per_launch": 4}, {"steps_per_launch": 4, "fused_z_chunk": 3}, {"steps_per_launch": 4, "fused_z_chunk": 16}):
                    got, kn = run_bits(shape, bits, t, w, opts)
                    ok = np.array_equal(got, exp)
                    if not ok:
                        bad += 1
                        d = np.argwhere(got != exp)
                        print("MISMATCH", dims, t, opts, kn, len(d), "cells; first", d[:4].tolist(), "z range", d[:, 0].min(), d[:, 0].max(),
                              "y", d[:, 1].min(), d[:, 1].max(), "x", d[:, 2].min(), d[:, 2].max(), flush=True)
            print("small", dims, "done; kernel", kn, "mismatches so far", bad, flush=True)
        big = [(70, 130, 248), (96, 200, 384)] if args.quick else [(70, 130, 248), (96, 200, 384), (200, 300, 520), (130, 768, 768)]
        for dims in big:
            bits = O.to_bf16(rng.standard_normal(O.padded_shape(shape, dims)))
            for t in (4, 8, 10):
                exp, _ = run_bits(shape, bits, t, w, {"steps_per_launch": 1})
                for opts in ({"steps_per_launch": 4}, {"steps_per_launch": 4, "fused_z_chunk": 24}):
                    got, kn = run_bits(shape, bits, t, w, opts)
                    if not np.array_equal(got, exp):
                        bad += 1
                        d = np.argwhere(got != exp)
                        print("MISMATCH (vs single sweeps)", dims, t, opts, kn, len(d), "cells; first", d[:4].tolist(), "z", d[:, 0].min(), d[:, 0].max(),
                              "y", d[:, 1].min(), d[:, 1].max(), "x", d[:, 2].min(), d[:, 2].max(), flush=True)
            print("big", dims, "done; kernel", kn, "mismatches so far", bad, flush=True)
        print("PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
    # timing: microseconds per launch at 768^3 (and a thin slab)
    for dims in ((768, 768, 768), (96, 768, 768)):
        ps = L.padded_shape(shape, dims)
        gen = torch.Generator(device="cuda").manual_seed(7)
        src = (torch.rand(ps, generator=gen, device="cuda") * 2 - 1).to(torch.bfloat16)
        dst = src.clone()
        pts = dims[0] * dims[1] * dims[2]
        for name, opts in (("lanes k=4", {"steps_per_launch": 4}), ("fused2 (r03 default)", {"lanes3": 0}),
                           ("lanes k=2", {"steps_per_launch": 4, "k2": 1})):
            plan = L.Plan(shape, dims, dtype="bf16").set_weights(w)
            k2 = opts.pop("k2", 0)
            for k, v in opts.items():
                plan.set_option(k, v)
            apps = plan.get_option("steps_per_launch")
            fn = (lambda: plan.step2(src, dst)) if k2 else (lambda: plan.stepk(src, dst))
            if k2:
                apps = 2
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            n = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(f"{dims} {name:22s} kernel {plan.kernel_name:30s} apps {apps} {us:9.1f} us/launch {pts * apps / us / 1e3:8.1f} GStencils/s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    t0 = time.time()
    rc = main()
    print("elapsed", round(time.time() - t0, 1), "s")
    sys.exit(rc)
