#!/usr/bin/env python3
"""Development tool: the workgroup-row 2D kernel (kernels_2d_wg.hip, six applications per launch) against the oracle
on small / ragged / rim-heavy grids, then its timing on the BASELINE grids beside the row-streaming kernel (four per
launch).  Writes gpurun_out/wg_check.jsonl."""
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


def parity(K=6, base=None):
    rng = np.random.default_rng(11)
    bad = 0
    cases = []
    for shape in ("star2d1r", "box2d3r", "star2d3r"):
        for dims in ((64, 128), (100, 250), (37, 476), (301, 1000), (8, 2), (1, 476), (13, 952), (700, 110), (40, 2100),
                     (333, 477), (50, 1429), (1200, 600)):
            for opts in ({}, {"wg_rows": 8}, {"wg_rows": 90}):
                cases.append((shape, dims, opts))
    for shape, dims, opts in cases:
        for bc in ("reference",):
            if bc == "dirichlet" and (opts or shape != "star2d1r" or dims[1] % 2):
                continue
            w = L.effective_weights(shape)
            w = w / w.sum()
            ps = L.padded_shape(shape, dims)
            a = rng.standard_normal(ps)
            plan = L.Plan(shape, dims).set_weights(w)
            if bc == "dirichlet":
                plan.set_boundary(bc)
            plan.set_option("stream", 1).set_option("steps_per_launch", K)
            for k, v in (base or {}).items():
                plan.set_option(k, v)
            for k, v in opts.items():
                plan.set_option(k, v)
            assert plan.get_option("steps_per_launch") == K, plan.get_option("steps_per_launch")
            src = torch.from_numpy(a).cuda()
            dst = torch.from_numpy(a).cuda()
            dst[4:-4, 4:-4] = -7.0
            plan.stepk(src, dst)
            torch.cuda.synchronize()
            got = dst.cpu().numpy()
            if bc == "reference":
                ref = O.run(shape, a, K, weights=w)
            else:
                ref = O.run_bc(shape, a, K, bc, weights=w)
            err = np.abs(got - ref).max()
            ok = err < 1e-12 and np.array_equal(got[:4], a[:4]) and np.array_equal(got[:, :4], a[:, :4])
            bad += not ok
            print(("ok  " if ok else "FAIL"), f"K={K}", shape, dims, opts, bc, f"err {err:.2e}", flush=True)
    # integer data: bit-exact against the oracle
    for shape in ("star2d1r", "star2d3r"):
        for dims in ((64, 128), (500, 1000), (97, 3)):
            a = O.reference_input(shape, dims)
            plan = L.Plan(shape, dims)
            plan.set_option("stream", 1).set_option("steps_per_launch", K)
            src = torch.from_numpy(a).cuda()
            dst = torch.from_numpy(a).cuda()
            plan.stepk(src, dst)
            torch.cuda.synchronize()
            got = dst.cpu().numpy()
            ref = O.run(shape, a, K)
            ok = np.array_equal(got, ref)
            bad += not ok
            print(("ok  " if ok else "FAIL"), f"K={K} exact", shape, dims, flush=True)
    return bad


def timing(out, iters):
    for shape, dims in (("star2d1r", (16384, 16384)),):
        w = L.effective_weights(shape)
        w = w / w.sum()
        ps = L.padded_shape(shape, dims)
        src = torch.rand(ps, dtype=torch.float64, device="cuda")
        dst = torch.zeros(ps, dtype=torch.float64, device="cuda")
        pts = dims[0] * dims[1]
        for K, opts in ((4, {}), (6, {}), (4, {"wg": 1}), (2, {"wg": 1}), (2, {}), (4, {"wg": 1, "wg_prio": 0}), (4, {"wg": 1, "wg_rows": 400}), (2, {"wg": 1, "wg_rows": 400}), (4, {})):
            plan = L.Plan(shape, dims).set_weights(w)
            plan.set_option("stream", 1).set_option("steps_per_launch", K)
            for k, v in opts.items():
                plan.set_option(k, v)
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
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    bad = 0
    if not args.no_parity:
        bad = parity() + parity(4, {"wg": 1}) + parity(2, {"wg": 1})
        print("parity mismatches:", bad, flush=True)
    if not args.no_timing and bad == 0:
        with open(os.path.join(ROOT, "gpurun_out", "wg_check.jsonl"), "a") as out:
            timing(out, args.iters)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
