#!/usr/bin/env python3
"""One rank's share of an N-GPU run, exchange included, on ONE GPU: the share (1/N of the outermost extent) runs as a
RING OF ONE over the real RCCL backend -- ghost zones refreshed by P2P messages the rank sends to itself, boundary strips
first, interior overlapped -- so N x its rate is what N GPUs would deliver if xGMI were as fast as a device-local copy: every
cost but the link.  A PROJECTION (no multi-GPU node in this pool).  2D: star2d1r 16384^2 (BASELINE config 2); 3D: star3d1r
512^3 fp64 and box3d1r 768^3 fp64 / bf16 (configs 4 / 5).  Writes gpurun_out/slab_shares.jsonl."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import lorastencil_amd as L
from lorastencil_amd import slab

def best(fn, n=3):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
os.environ["LORA_SLAB_EXCHANGE"] = "p2p"
out = open(os.path.join(ROOT, "gpurun_out", "slab_shares.jsonl"), "a")
cases = []
for ngpu in (1, 2, 4, 8):
    for opts in ({}, {"steps_per_launch": 6}, {"wg": 1, "steps_per_launch": 4}):
        cases.append(("star2d1r", (16384, 16384), "f64", ngpu, opts, 96))
    for opts in ({}, {"steps_per_launch": 2}):
        cases.append(("star3d1r", (512, 512, 512), "f64", ngpu, opts, 48))
        cases.append(("box3d1r", (768, 768, 768), "f64", ngpu, opts, 48))
    cases.append(("box3d1r", (768, 768, 768), "bf16", ngpu, {}, 48))
for shape, gdims, dtype, ngpu, opts, steps in cases:
    dims = (gdims[0] // ngpu,) + tuple(gdims[1:])
    w = L.effective_weights(shape)
    w = w / w.sum()
    try:
        ring = slab.SlabDriver(shape, dims, device="cuda:0", weights=w, dtype=dtype, ring_of_one=(ngpu > 1), options=opts)
    except Exception as e:  # noqa: BLE001
        print("skip", shape, dims, opts, e, flush=True)
        continue
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float64
    local = torch.rand(ring.local_padded_shape, device="cuda").to(tdt)
    def run():
        ring.load_local(local); ring.refresh_ghosts(); ring.run(steps)
    def load():
        ring.load_local(local); ring.refresh_ghosts()
    run()
    tr, tl = best(run), best(load)
    pts = 1
    for d in dims:
        pts *= d
    rate = pts * steps / (tr - tl) / 1e9
    rec = {"shape": shape, "global": gdims, "dtype": dtype, "gpus": ngpu, "share": dims, "options": opts, "kernel": ring.stepper.plan.kernel_name,
           "apps_per_launch": ring.apps, "ghost": ring.layout.ghost, "exchange_every": ring.exchange_every,
           "gstencils_per_rank": round(rate, 1), "projected_gstencils": round(rate * ngpu, 1)}
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n"); out.flush()
    del ring, local
    torch.cuda.empty_cache()
dist.destroy_process_group()
