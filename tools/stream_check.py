#!/usr/bin/env python3
"""Development tool: the row-streaming fused 2D kernel (option stream=1) against the tile kernel (bit for bit) and the
oracle on ragged sizes, then a timing sweep of its knobs on the BASELINE grids.  Writes gpurun_out/stream_check.jsonl."""
import argparse
import itertools
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
    rng = np.random.default_rng(7)
    bad = 0
    for shape in ("star2d1r", "box2d3r", "star2d3r"):
        for dims in ((64, 128), (100, 250), (37, 118), (301, 1000), (8, 2), (1, 116), (13, 232), (700, 120)):
            for opts in ({}, {"stream_rows": 8}, {"stream_rows": 50, "stream_depth": 6, "stream_sync": 0},
                         {"stream_depth": 2}, {"lowrank_valu": 0, "stream_depth": 3}):
                w = L.effective_weights(shape)
                w = w / w.sum()
                ps = L.padded_shape(shape, dims)
                a = rng.standard_normal(ps)
                src = torch.from_numpy(a).cuda()
                outs = []
                for stream in (0, 1):
                    plan = L.Plan(shape, dims).set_weights(w)
                    plan.set_option("stream", stream)
                    if stream:
                        for k, v in opts.items():
                            plan.set_option(k, v)
                    elif "lowrank_valu" in opts:
                        plan.set_option("lowrank_valu", opts["lowrank_valu"])
                    dst = torch.full(ps, -7.0, dtype=torch.float64, device="cuda")
                    plan.step2(src, dst)
                    torch.cuda.synchronize()
                    outs.append(dst.cpu().numpy())
                same = np.array_equal(outs[0], outs[1])
                # oracle: two sweeps with a zero halo in between
                mid = np.zeros_like(a)
                O.step(shape, a, w, out=mid)
                ref = np.full_like(a, -7.0)
                O.step(shape, mid, w, out=ref)
                err = np.abs(outs[1] - ref).max()
                ok = same and err < 1e-12
                bad += not ok
                print(("ok  " if ok else "FAIL"), shape, dims, opts, "identical" if same else "DIFFERENT", f"err {err:.2e}", flush=True)
                # region launch: rows [b, e) only
                if dims[0] >= 37:
                    b, e = 5, dims[0] - 9
                    plan = L.Plan(shape, dims).set_weights(w)
                    plan.set_option("stream", 1)
                    for k, v in opts.items():
                        plan.set_option(k, v)
                    dst = torch.full(ps, -7.0, dtype=torch.float64, device="cuda")
                    plan.step2_region(src, dst, b, e)
                    torch.cuda.synchronize()
                    got = dst.cpu().numpy()
                    exp = np.full_like(a, -7.0)
                    exp[4 + b:4 + e, 4:-4] = outs[0][4 + b:4 + e, 4:-4]
                    if not np.array_equal(got, exp):
                        bad += 1
                        print("FAIL region", shape, dims, opts, flush=True)
    # K = 4: one launch = four sweeps of the reference driver from an even level (O.run keeps the alternating halo)
    for shape in ("star2d1r", "box2d3r", "star2d3r"):
        for dims in ((64, 128), (100, 250), (37, 104), (301, 1000), (8, 2), (1, 104), (13, 208), (700, 110), (40, 2100)):
            for opts in ({}, {"stream_rows": 8}, {"stream_rows": 50, "stream_depth": 2, "stream_sync": 0},
                         {"stream_sync": 2}, {"lowrank_valu": 0}):
                w = L.effective_weights(shape)
                w = w / w.sum()
                ps = L.padded_shape(shape, dims)
                a = rng.standard_normal(ps)
                plan = L.Plan(shape, dims).set_weights(w)
                plan.set_option("stream", 1).set_option("steps_per_launch", 4)
                for k, v in opts.items():
                    plan.set_option(k, v)
                assert plan.get_option("steps_per_launch") == 4
                src = torch.from_numpy(a).cuda()
                dst = torch.from_numpy(a).cuda()
                dst[4:-4, 4:-4] = -7.0
                plan.stepk(src, dst)
                torch.cuda.synchronize()
                got = dst.cpu().numpy()
                ref = O.run(shape, a, 4, weights=w)
                err = np.abs(got - ref).max()
                ok = err < 1e-12 and np.array_equal(got[:4], a[:4]) and np.array_equal(got[:, :4], a[:, :4])
                bad += not ok
                print(("ok  " if ok else "FAIL"), "K=4", shape, dims, opts, f"err {err:.2e}", flush=True)
            # the time-step driver with the mixed 4 / 2 / 1 schedule, halo included
            a = O.reference_input(shape, dims)
            w = L.effective_weights(shape)
            w = w / w.sum()
            for times in (4, 5, 6, 7, 8, 9, 10, 11, 13):
                plan = L.Plan(shape, dims).set_weights(w)
                plan.set_option("stream", 1).set_option("steps_per_launch", 4)
                b0 = torch.from_numpy(a).cuda()
                b1 = torch.zeros_like(b0)
                plan.run(b0, b1, times)
                torch.cuda.synchronize()
                got = (b0, b1)[times % 2].cpu().numpy()
                ref = O.run(shape, a, times, weights=w)
                ok = np.abs(got - ref).max() < 1e-11 and np.array_equal(got[:4], ref[:4]) and np.array_equal(got[:, -4:], ref[:, -4:])
                bad += not ok
                print(("ok  " if ok else "FAIL"), "run K=4", shape, dims, times, flush=True)
    # Dirichlet: plan.run with the boundary option, stream vs tile
    for shape in ("star2d1r", "box2d3r"):
        for dims in ((64, 128), (100, 250), (301, 1000)):
            w = L.effective_weights(shape)
            w = w / w.sum()
            ps = L.padded_shape(shape, dims)
            a = rng.standard_normal(ps)
            res = []
            for stream in (0, 1):
                plan = L.Plan(shape, dims).set_weights(w).set_boundary("dirichlet")
                plan.set_option("stream", stream)
                b0 = torch.from_numpy(a).cuda()
                b1 = torch.zeros_like(b0)
                plan.run(b0, b1, 6)
                torch.cuda.synchronize()
                res.append(b0.cpu().numpy())
            ok = np.array_equal(res[0], res[1])
            bad += not ok
            print(("ok  " if ok else "FAIL"), "dirichlet", shape, dims, flush=True)
    print("parity failures:", bad, flush=True)
    return bad


def timing(out_path, quick):
    dev = torch.device("cuda", 0)

    def record(**kw):
        print(json.dumps(kw), flush=True)
        with open(out_path, "a") as f:
            f.write(json.dumps(kw) + "\n")

    a = torch.empty((16392, 16392), dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    t = time_fn(lambda: b.copy_(a), 20)
    record(kind="copy", bytes=2 * a.numel() * 8, seconds=t, gbs=2 * a.numel() * 8 / t / 1e9)
    del a, b
    cases = [("star2d1r", (16384, 16384)), ("box2d3r", (8192, 8192)), ("star2d3r", (16384, 16384))]
    for shape, dims in cases:
        w = L.effective_weights(shape)
        w = w / w.sum()
        ps = L.padded_shape(shape, dims)
        src = torch.randint(0, 100, ps, device=dev).to(torch.float64)
        dst = torch.zeros_like(src)
        pts = dims[0] * dims[1]
        grid = [{"stream": 0, "steps_per_launch": 2}, {"steps_per_launch": 1}, {"steps_per_launch": 2}]
        for lr in ([-1, 4] if shape == "star2d1r" else [-1]):
            for r in ([0, 290, 330, 580, 870] if not quick else [0]):
                grid.append({"steps_per_launch": 4, "stream_rows": r, "lowrank_valu": lr})
        for opts in grid:
            plan = L.Plan(shape, dims).set_weights(w)
            for k, v in opts.items():
                plan.set_option(k, v)
            apps = plan.get_option("steps_per_launch")
            t = time_fn(lambda: (plan.stepk(src, dst), plan.stepk(dst, src)), 10) / 2
            record(kind="stepk", shape=shape, dims=dims, opts=opts, kernel=plan.kernel_signature, us=round(t * 1e6, 1),
                   gstencils=round(apps * pts / t / 1e9, 1), real_tbs=round(2 * pts * 8 / t / 1e12, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stream_check.jsonl"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    bad = 0
    if not args.no_parity:
        bad = parity()
        if bad:
            sys.exit(1)  # do not time a wrong kernel
    if not args.no_timing:
        timing(args.out, args.quick)


if __name__ == "__main__":
    main()
