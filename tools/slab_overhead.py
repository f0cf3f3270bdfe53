#!/usr/bin/env python3
"""Host overhead of the Python slab driver: a 2048 x 16384 slab (the 8-GPU share of the headline grid) swept 100 times
through SlabDriver (one launch per call) vs lora_plan_run (one C call), single rank, no exchange -- and then the same
slab as a RING OF ONE over the real RCCL backend: ghost zones of 24 rows per side refreshed every 4 launches by P2P
messages the rank sends to itself (boundary strips first, interior overlapped), i.e. one rank's share of the 8-GPU run
including its exchange, on one GPU.  8 x that rate is what 8 GPUs would deliver if xGMI were as fast as a device-local
copy; it isolates every cost except the link itself."""
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
      f"overhead per launch: {(ts-tp)/25*1e6:.1f} us")
# pure host cost of issuing the launches (GPU not the bottleneck: tiny grid)
small = slab.SlabDriver("star2d1r", (64, 128), device="cuda:0")
small.load_local(torch.zeros(small.local_padded_shape, dtype=torch.float64, device="cuda"))
th = t(lambda: small.run(1000))
print(f"host cost per slab launch (tiny grid): {th/500*1e6:.1f} us")

# one rank's share of the N-GPU run, exchange included, over the real RCCL backend (the rank is its own neighbour)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29590")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cases = [(8192, 2, "p2p", None, True), (4096, 4, "p2p", None, True), (2048, 8, "p2p", None, True),
         (2048, 8, "p2p", 1, True), (2048, 8, "p2p", 2, True), (2048, 8, "p2p", 4, True), (2048, 8, "p2p", 16, True),
         (2048, 8, "p2p", None, False), (2048, 8, "allgather", None, True)]
for rows, ngpu, mode, every, overlap in cases:
    d = (rows, 16384)
    srcr = torch.randint(0, 100, (rows + 8, 16392), device="cuda").to(torch.float64)
    if True:
        os.environ["LORA_SLAB_EXCHANGE"] = mode
        ring = slab.SlabDriver("star2d1r", d, device="cuda:0", weights=w, ring_of_one=True, exchange_every=every,
                               overlap=overlap)
        local = torch.zeros(ring.local_padded_shape, dtype=torch.float64, device="cuda")
        g = ring.layout.ghost
        local[4 + g:4 + g + rows] = srcr[4:4 + rows]
        def run_ring():
            ring.load_local(local); ring.refresh_ghosts(); ring.run(100)
        tr = t(run_ring)
        def run_load():
            ring.load_local(local); ring.refresh_ghosts()
        tl = t(run_load)
        rate = rows * 16384 * 100 / (tr - tl) / 1e9
        print(f"ring of one, {rows} x 16384 ({ngpu}-GPU share), {ring.apps} applications per launch, ghost {g} (every {ring.exchange_every} launches), {mode}, "
              f"overlap {overlap}: {rate:.0f} GSt/s per rank -> x{ngpu} = {rate * ngpu:.0f} GSt/s if the links kept up "
              f"({(tr - tl) * 1e3 / 100:.4f} ms per sweep)")
dist.destroy_process_group()
