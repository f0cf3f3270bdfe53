#!/usr/bin/env python3
"""Development tool: every workgroup shape of the 3D plane-streaming kernel on a grid large enough for many workgroups
per CU and several rounds (races between waves show only at scale), against single sweeps, bit for bit, repeated."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402

bad = n = 0
for shape, dims in (("star3d1r", (384, 500, 616)), ("box3d1r", (300, 384, 480))):
    w = L.effective_weights(shape)
    w = w / w.sum()
    ps = L.padded_shape(shape, dims)
    torch.manual_seed(3)
    a = torch.randn(ps, dtype=torch.float64, device="cuda")

    def run(opts, times):
        plan = L.Plan(shape, dims).set_weights(w)
        if "stream3" not in opts and opts.get("steps_per_launch") != 1:
            plan.set_option("stream3", 1)  # the plane-streaming kernel whatever the grid size
        for k, v in opts.items():
            plan.set_option(k, v)
        b0 = a.clone()
        b1 = torch.zeros_like(b0)
        plan.run(b0, b1, times)
        torch.cuda.synchronize()
        return (b0, b1)[times % 2], plan.kernel_signature

    for times in (6, 7):
        ref, _ = run({"steps_per_launch": 1}, times)
        for k in (3, 2):
            for wv in (8, 4):
                for extra in ({}, {"fused_z_chunk": 32}, {"stream3_pipe": 1}, {"stream3_async": 1}):
                    opts = dict({"steps_per_launch": k, "stream3_waves": wv}, **extra)
                    for rep in range(3):
                        got, sig = run(opts, times)
                        n += 1
                        same = torch.equal(got, ref)
                        if not same and "taps=2" in sig:
                            # separable x / y / z evaluation of the box's taps: another summation order
                            same = float((got - ref).abs().max()) <= 1e-13 * float(ref.abs().max())
                        if not same:
                            bad += 1
                            print("MISMATCH", shape, times, sig, "rep", rep, int((got != ref).sum()), "cells", flush=True)
        print(shape, times, "done", flush=True)
print(f"stress: {n - bad}/{n} runs bit-identical to single sweeps")
sys.exit(1 if bad else 0)
