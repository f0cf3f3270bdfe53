#!/usr/bin/env python3
"""Ghost-zone refresh interval E of the slab driver on the six-sweep 2D kernel (and the 3D four-sweep one): one rank's share
of an N-GPU run as a RING OF ONE over RCCL (tools/slab_shares.py), E launches between refreshes, ghost = radius x K x E rows
either side -- all of them swept by every launch.  Writes gpurun_out/slab_every_sweep.jsonl."""
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
os.environ.setdefault("MASTER_PORT", "29593")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
os.environ["LORA_SLAB_EXCHANGE"] = "p2p"
out = open(os.path.join(ROOT, "gpurun_out", "slab_every_sweep.jsonl"), "a")
cases = []
for ngpu in (8, 4, 2):
    for e in (None, 1, 2, 3, 4, 8):
        cases.append(("star2d1r", (16384, 16384), ngpu, e, 96))
for ngpu in (8, 4):
    for e in (None, 1, 2, 4):
        cases.append(("star3d1r", (512, 512, 512), ngpu, e, 48))
        cases.append(("box3d1r", (768, 768, 768), ngpu, e, 48))
for shape, gdims, ngpu, e, steps in cases:
    dims = (gdims[0] // ngpu,) + tuple(gdims[1:])
    w = L.effective_weights(shape)
    w = w / w.sum()
    ring = slab.SlabDriver(shape, dims, device="cuda:0", weights=w, ring_of_one=True, exchange_every=e)
    local = torch.rand(ring.local_padded_shape, device="cuda", dtype=torch.float64)
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
    rec = {"shape": shape, "gpus": ngpu, "share": dims, "requested_every": e, "exchange_every": ring.exchange_every,
           "apps_per_launch": ring.apps, "ghost": ring.layout.ghost, "gstencils_per_rank": round(rate, 1),
           "projected_gstencils": round(rate * ngpu, 1)}
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n"); out.flush()
    del ring, local
    torch.cuda.empty_cache()
dist.destroy_process_group()
