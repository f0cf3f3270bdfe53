#!/usr/bin/env python3
"""Development tool: the register-resident 3D kernel (kernels_3d_lanes.hip, four applications per launch) against the
oracle on small / ragged / rim-heavy / odd grids, then its timing on the BASELINE grids beside the plane-streaming
kernel (three per launch).  Writes gpurun_out/lanes3_check.jsonl."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)


def time_fn(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


def parity():
    rng = np.random.default_rng(5)
    bad = 0
    for shape in ("star3d1r", "box3d1r"):
        for dims in ((8, 16, 128), (30, 24, 120), (41, 50, 250), (3, 5, 7), (1, 1, 2), (70, 100, 260), (12, 24, 121),
                     (33, 47, 255), (9, 200, 64), (100, 25, 130)):
            for opts in ({}, {"fused_z_chunk": 5}, {"fused_z_chunk": 33}):
                w = L.effective_weights(shape)
                w = w / w.sum()
                ps = L.padded_shape(shape, dims)
                a = rng.standard_normal(ps)
                plan = L.Plan(shape, dims).set_weights(w)
                plan.set_option("steps_per_launch", 4)
                for k, v in opts.items():
                    plan.set_option(k, v)
                assert plan.get_option("steps_per_launch") == 4 and plan.kernel_name == "stencil3d_lanes_kernel", plan.kernel_name
                src = torch.from_numpy(a).cuda()
                dst = torch.from_numpy(a).cuda()
                dst[1:-1, 2:-2, 4:-4] = -7.0
                plan.stepk(src, dst)
                torch.cuda.synchronize()
                got = dst.cpu().numpy()
                ref = O.run(shape, a, 4, weights=w)
                err = np.abs(got - ref).max()
                ok = err < 1e-12 and np.array_equal(got[:1], a[:1]) and np.array_equal(got[:, :2], a[:, :2]) and np.array_equal(got[:, :, :4], a[:, :, :4])
                bad += not ok
                print(("ok  " if ok else "FAIL"), "K=4", shape, dims, opts, f"err {err:.2e}", flush=True)
            # the time-step driver: 4-launches, the two-application tail, single sweeps; whole padded buffers
            a = O.reference_input(shape, dims)
            for times in (4, 5, 6, 7, 8, 9, 12, 13):
                plan = L.Plan(shape, dims).set_option("steps_per_launch", 4)
                b0 = torch.from_numpy(a).cuda()
                b1 = torch.zeros_like(b0)
                plan.run(b0, b1, times)
                torch.cuda.synchronize()
                got = (b0, b1)[times % 2].cpu().numpy()
                ref = O.run(shape, a, times)
                ok = np.array_equal(got, ref) if np.abs(ref).max() < 2.0 ** 50 else np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
                bad += not ok
                print(("ok  " if ok else "FAIL"), "run", shape, dims, times, flush=True)
    return bad


def timing(out, iters):
    for shape, dims in (("star3d1r", (512, 512, 512)), ("box3d1r", (768, 768, 768)), ("star3d1r", (768, 768, 768)),
                        ("star3d1r", (256, 256, 256)), ("star3d1r", (384, 384, 384)), ("box3d1r", (512, 512, 512)), ("star3d1r", (128, 128, 128)),
                        ("star3d1r", (192, 192, 192)), ("box3d1r", (256, 256, 256)), ("star3d1r", (512, 512, 511))):
        w = L.effective_weights(shape)
        w = w / w.sum()
        ps = L.padded_shape(shape, dims)
        src = torch.rand(ps, dtype=torch.float64, device="cuda")
        dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
        pts = dims[0] * dims[1] * dims[2]
        for opts in ({"lanes3": 0}, {"steps_per_launch": 4}, {"steps_per_launch": 4, "fused_z_chunk": 48}, {"steps_per_launch": 4, "fused_z_chunk": 96}):
            plan = L.Plan(shape, dims).set_weights(w)
            for k, v in opts.items():
                plan.set_option(k, v)
            K = plan.get_option("steps_per_launch")
            t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), iters) / 2
            rec = {"shape": shape, "dims": dims, "K": K, "opts": opts, "kernel": plan.kernel_name, "us_per_launch": t * 1e6,
                   "gstencils": pts * K / t / 1e9}
            print(json.dumps(rec), flush=True)
            out.write(json.dumps(rec) + "\n")
            out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    bad = 0
    if not args.no_parity:
        bad = parity()
        print("parity mismatches:", bad, flush=True)
    if not args.no_timing and bad == 0:
        with open(os.path.join(ROOT, "gpurun_out", "lanes3_check.jsonl"), "a") as out:
            timing(out, args.iters)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
