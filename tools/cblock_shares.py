#!/usr/bin/env python3
"""Slabs against two-axis blocks through the C++ drivers, all eight ranks' shares run one after the other on ONE GPU with
the loopback exchange (every cost of the decomposition but the link): aggregate GStencils/s, redundant cells, bytes per
rank and refresh.  star2d1r 16384^2, star3d1r 512^3, box3d1r 768^3 fp64 and bf16.
    python tools/cblock_shares.py > gpurun_out/cblock_shares.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lorastencil_amd as L  # noqa: E402
from lorastencil_amd import cblocks  # noqa: E402


def best(fn, n=3):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return b


cases = [("star2d1r", (16384, 16384), "f64", 48), ("star3d1r", (512, 512, 512), "f64", 48), ("box3d1r", (768, 768, 768), "f64", 24),
         ("box3d1r", (768, 768, 768), "bf16", 48)]
for shape, dims, dtype, steps in cases:
    w = L.effective_weights(shape)
    w = w / w.sum()
    rng = np.random.default_rng(1)
    a = rng.random(L.padded_shape(shape, dims))
    if dtype == "bf16":
        a = L.to_bf16(a) if hasattr(L, "to_bf16") else (a.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    pts = 1
    for d in dims:
        pts *= d
    for grid in ((1, 1), (8, 1), (4, 2), (2, 4)):
        n = grid[0] * grid[1]
        for every in ((1, 2) if n > 1 else (1,)):
            g = cblocks.BlockGrid(shape, dims, grid, comms=cblocks.loopback_comms(n) if n > 1 else None, dtype=dtype, weights=w, exchange_every=every)
            g.load(a)

            def run():
                g.run(steps)
                g.sync()
            run()
            t = best(run)
            i0 = g.info(n // 2)  # a middle block
            loc = 1
            for k in range(len(dims)):
                loc *= i0.local_dims[k]
            own = pts / n
            rec = {"driver": "csrc/blocks.cpp (loopback, one GPU)", "shape": shape, "dtype": dtype, "dims": dims, "grid": grid,
                   "apps_per_launch": i0.apps_per_launch, "exchange_every": i0.exchange_every, "ghost": i0.ghost,
                   "redundant_cells_pct": round(100.0 * (loc / own - 1), 1), "bytes_per_rank_per_refresh": i0.bytes_per_refresh,
                   "aggregate_gstencils": round(pts * steps / t / 1e9, 1)}
            print(json.dumps(rec), flush=True)
            g.close()
            del g
            torch.cuda.empty_cache()
