#!/usr/bin/env python3
"""One rank over the real "nccl" (= RCCL) backend sending to ITSELF: exercises the point-to-point path the slab driver
uses (batch_isend_irecv on row slices of device tensors, stream ordering after a kernel launch, Work.wait) on a 1-GPU
box, plus the all-gather fallback.  Development tool; prints RCCL_SELF_P2P_OK or the error."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29588")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    x = torch.zeros((64, 4100), dtype=torch.float64, device="cuda")
    x[8:16] = torch.arange(8 * 4100, dtype=torch.float64, device="cuda").view(8, 4100)  # producer kernel, same stream
    ops = [dist.P2POp(dist.isend, x[8:16], 0), dist.P2POp(dist.irecv, x[40:48], 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    y = x[40:48] * 2.0  # consumer kernel on the current stream, ordered after the receive by wait()
    torch.cuda.synchronize()
    ok = torch.equal(x[40:48], x[8:16]) and torch.equal(y, x[8:16] * 2.0)
    mine = x[8:16].contiguous()
    flat = torch.empty_like(mine)
    dist.all_gather_into_tensor(flat, mine, async_op=True).wait()
    torch.cuda.synchronize()
    ok = ok and torch.equal(flat, mine)
    print("RCCL_SELF_P2P_OK" if ok else "RCCL_SELF_P2P_MISMATCH", flush=True)
except Exception as e:  # noqa: BLE001
    print("RCCL_SELF_P2P_ERROR", type(e).__name__, str(e)[:500], flush=True)
finally:
    dist.destroy_process_group()
