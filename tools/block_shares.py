#!/usr/bin/env python3
"""2-D blocks against row slabs on ONE GPU (loopback exchanges): star2d1r 16384^2 cut into 8 shares either as 8 x 1 row
slabs (csrc/slab.cpp, the default decomposition) or as a 2 x 4 / 4 x 2 block grid (lorastencil_amd/blocks.py).  All shares
run one after the other on this device, so the aggregate rate is what ONE GPU delivers on the decomposed problem: its
ratio to the undivided run is the decomposition's compute + copy overhead (redundant ghost cells, narrower strips, pack /
unpack kernels), every cost but the link.  Also prints the per-rank bytes each layout would put on xGMI.
Writes gpurun_out/block_shares.jsonl."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lorastencil_amd as L
from lorastencil_amd import blocks, cslab

shape, dims, steps = "star2d1r", (16384, 16384), 96
out = open(os.path.join(ROOT, "gpurun_out", "block_shares.jsonl"), "a")

def best(fn, n=3):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b

def emit(rec):
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n"); out.flush()

pts = dims[0] * dims[1]
a = np.random.default_rng(1).random((dims[0] + 8, dims[1] + 8))
a *= 0.01  # with the weights normalised below the run stays finite
w = L.effective_weights(shape)
w = w / w.sum()

# undivided
one = cslab.SlabSet(shape, dims, 1, weights=w)
one.load(a)
def run_one():
    one.run(steps); one.sync()
run_one()
t = best(run_one)
emit({"layout": "undivided", "gstencils": round(pts * steps / t / 1e9, 1)})
one.close()

for every in (1, 2):
    s8 = cslab.SlabSet(shape, dims, 8, comms=cslab.loopback_comms(8), exchange_every=every, weights=w)
    s8.load(a)
    def run_s():
        s8.run(steps); s8.sync()
    run_s()
    t = best(run_s)
    si = s8.info(1)
    emit({"layout": "8 x 1 slabs", "exchange_every": si.exchange_every, "ghost": si.ghost, "gstencils": round(pts * steps / t / 1e9, 1),
          "xgmi_bytes_per_rank_per_exchange": 2 * si.ghost * (dims[1] + 8) * 8,
          "redundant_cells": round((dims[0] // 8 + 2 * si.ghost) / (dims[0] // 8) - 1, 4)})
    s8.close()

for grid in ((2, 4), (4, 2)):
    for every in (1, 2):
        bs = blocks.BlockSet(shape, dims, grid, weights=w, exchange_every=every)
        bs.load(a)
        def run_b():
            bs.run(steps)
        run_b()
        t = best(run_b)
        g = bs.ghost
        own = (dims[0] // grid[0], dims[1] // grid[1])
        inner = blocks.BlockLayout(dims, grid, (min(1, grid[0] - 1), min(1, grid[1] - 1)), g)
        emit({"layout": f"{grid[0]} x {grid[1]} blocks", "exchange_every": bs.exchange_every, "ghost": g,
              "gstencils": round(pts * steps / t / 1e9, 1), "exchanges": bs.exchanges,
              "xgmi_bytes_per_rank_per_exchange": 8 * (g * own[0] * ((inner.gl > 0) + (inner.gr > 0)) + g * (inner.padded[1]) * ((inner.gt > 0) + (inner.gb > 0))),
              "redundant_cells": round(inner.local_dims[0] * inner.local_dims[1] / (own[0] * own[1]) - 1, 4)})
        del bs
        torch.cuda.empty_cache()
