#!/usr/bin/env python3
"""Does the fused star2d1r launch depend on the DATA (power / clocks)?  Times lora_plan_step2 on zeros, small integers
and full-mantissa values, same grid, same process (development tool; prints JSON lines)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402


def main():
    shape, dims = "star2d1r", (16384, 16384)
    plan = L.Plan(shape, dims)
    w = plan.weights
    plan.set_weights(w / w.sum())
    ps = plan.padded_shape
    dev = torch.device("cuda", 0)
    dst = torch.zeros(ps, dtype=torch.float64, device=dev)
    pts = dims[0] * dims[1]
    for name, make in (("zeros", lambda: torch.zeros(ps, dtype=torch.float64, device=dev)),
                       ("ints_0_99", lambda: torch.randint(0, 100, ps, device=dev).to(torch.float64)),
                       ("normal", lambda: torch.empty(ps, dtype=torch.float64, device=dev).normal_()),
                       ("zeros_again", lambda: torch.zeros(ps, dtype=torch.float64, device=dev))):
        src = make()
        for _ in range(5):
            plan.step2(src, dst)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            plan.step2(src, dst)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 1e3 / 50
        print(json.dumps({"data": name, "ms_per_launch": round(t * 1e3, 4), "gstencils": round(2 * pts / t / 1e9, 1)}), flush=True)
        del src


if __name__ == "__main__":
    main()
