#!/usr/bin/env python3
"""Host overhead of the Python slab driver: a 2048 x 16384 slab (the 8-GPU share of the headline grid) swept 100 times
through SlabDriver (one launch per call) vs lora_plan_run (one C call), single rank, no exchange."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lorastencil_amd as L
from lorastencil_amd import slab

dims = (2048, 16384)
w = L.effective_weights("star2d1r") / 100.0
plan = L.Plan("star2d1r", dims).set_weights(w)
ps = plan.padded_shape
src = torch.randint(0, 100, ps, device="cuda").to(torch.float64)
b0, b1 = src.clone(), torch.zeros_like(src)
def t(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
tp = t(lambda: plan.run(b0, b1, 100))
drv = slab.SlabDriver("star2d1r", dims, device="cuda:0", weights=w)
drv.load_local(src)
ts = t(lambda: drv.run(100))
pts = dims[0] * dims[1] * 100
print(f"plan.run: {tp*1e3:.2f} ms ({pts/tp/1e9:.0f} GSt/s)   SlabDriver.run: {ts*1e3:.2f} ms ({pts/ts/1e9:.0f} GSt/s)   "
      f"overhead per launch: {(ts-tp)/50*1e6:.1f} us")
# pure host cost of issuing the launches (GPU not the bottleneck: tiny grid)
small = slab.SlabDriver("star2d1r", (64, 128), device="cuda:0")
small.load_local(torch.zeros(small.local_padded_shape, dtype=torch.float64, device="cuda"))
th = t(lambda: small.run(1000))
print(f"host cost per slab launch (tiny grid): {th/500*1e6:.1f} us")
