#!/usr/bin/env python3
"""Does the timed region of `bench.py` (star2d1r 16384^2, 5 warm-up sweeps, 100 timed) sometimes run several times slower,
and if it does, for how long?  One process: the bench's prelude, then the 100 sweeps timed five times in a row.
   python tools/slow_region_hunt.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402

shape, dims = "star2d1r", (16384, 16384)
w = L.effective_weights(shape)
w = w / w.sum()
plan = L.Plan(shape, dims).set_weights(w)
gen = torch.Generator(device="cuda").manual_seed(1234)
src0 = torch.randint(0, 100, plan.padded_shape, generator=gen, device="cuda").to(torch.float64)
b0, b1 = src0.clone(), torch.zeros_like(src0)
plan.run(b0, b1, 5)
torch.cuda.synchronize()
out = []
for rep in range(5):
    b0.copy_(src0)
    b1.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.run(b0, b1, 100)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) * 1e3, 2))
print("ms per 100 sweeps, five times in a row:", out, flush=True)
