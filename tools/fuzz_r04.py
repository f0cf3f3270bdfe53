#!/usr/bin/env python3
"""Development tool (round 4): randomized parity sweep of this round's launch paths on the GPU -- the register-resident 3D
kernels under every cut of a launch (chunks, spans, team spans, fixed chunks; chunks dealt to resident workgroups; the extra
chunk of the slow rim tiles; reduced-level first steps), fp64 and bf16, two ranges in one launch, the periodic option on a
ghost-extended grid, general 49-tap tables at six sweeps per launch -- each against single sweeps of the same plan family.
   python tools/fuzz_r04.py [--seconds 150] [--seed 1]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=150.0)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)


def pick(xs):
    return xs[int(rng.integers(len(xs)))]


def run(shape, dims, dtype, w, a, times, opts, boundary="reference"):
    plan = L.Plan(shape, dims, dtype=dtype).set_weights(w)
    if boundary != "reference":
        plan.set_boundary(boundary)
    for k, v in opts.items():
        plan.set_option(k, v)
    b0 = a.clone()
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, times)
    torch.cuda.synchronize()
    return (b0, b1)[times % 2], plan.kernel_signature


def bits(t):
    return t.view(torch.int16) if t.dtype == torch.bfloat16 else t.view(torch.int64)


def close(got, ref, tol):
    if torch.equal(bits(got), bits(ref)):
        return True
    if tol == 0:
        return False
    g, r = got.double(), ref.double()
    return float((g - r).abs().max()) <= tol * max(float(r.abs().max()), 1e-300)


t_end = time.time() + args.seconds
n = bad = 0
kinds = {}
while time.time() < t_end:
    kind = pick(["lanes", "lanes", "lanes_bf16", "torus", "general49", "ranges"])
    try:
        if kind in ("lanes", "lanes_bf16", "ranges"):
            bf16 = kind == "lanes_bf16" or (kind == "ranges" and rng.random() < 0.4)
            shape = "box3d1r" if bf16 else pick(["star3d1r", "box3d1r"])
            big = rng.random() < 0.35
            dims = (int(rng.integers(1, 260 if big else 50)), int(rng.integers(1, 420 if big else 90)),
                    (8 if bf16 else 2) * int(rng.integers(1, (80 if big else 20) if bf16 else (320 if big else 80))) + (0 if bf16 or rng.random() < 0.85 else 1))
            dtype = "bf16" if bf16 else "f64"
            tdt = torch.bfloat16 if bf16 else torch.float64
            w = L.effective_weights(shape)
            w = w / w.sum()
            a = torch.from_numpy(rng.standard_normal(L.padded_shape(shape, dims))).to(tdt).cuda()
            opts = {"steps_per_launch": 4, "spans3": int(pick([-1, -1, 0, 1, 2]))}
            if rng.random() < 0.3:
                opts["fused_z_chunk"] = int(pick([1, 3, 8, 24, 64]))
            tol = 0 if (bf16 or shape == "star3d1r") else 1e-13
            if kind == "ranges":
                plan = L.Plan(shape, dims, dtype=dtype).set_weights(w)
                for k, v in opts.items():
                    plan.set_option(k, v)
                napps = int(pick([4, 2]))
                h = dims[0]
                cuts = sorted(int(x) for x in rng.integers(0, h + 1, 2))
                whole = a.clone()
                plan.stepn_region(napps, a, whole, 0, h)
                got = a.clone()
                plan.stepn_region2(napps, a, got, 0, cuts[0], cuts[1], h)
                plan.stepn_region(napps, a, got, cuts[0], cuts[1])
                torch.cuda.synchronize()
                ok, what = close(got, whole, 0), (shape, dims, dtype, napps, cuts, opts, plan.kernel_signature)
            else:
                times = int(pick([4, 5, 6, 8, 9, 13]))
                ref, _ = run(shape, dims, dtype, w, a, times, {"steps_per_launch": 1})
                got, sig = run(shape, dims, dtype, w, a, times, opts)
                ok, what = close(got, ref, tol), (shape, dims, dtype, times, opts, sig)
        elif kind == "torus":
            shape = pick(["1d1r", "star2d1r", "box2d3r", "star2d3r", "star3d1r", "box3d1r", "box3d1r"])
            nd = 1 if shape.startswith("1d") else (2 if "2d" in shape else 3)
            bf16 = nd == 3 and shape == "box3d1r" and rng.random() < 0.5
            if nd == 1:
                dims = (int(pick([300, 1000, 4096, 65536])),)
            elif nd == 2:
                dims = (int(rng.integers(22, 400)), 2 * int(rng.integers(11, 300)) + int(rng.random() < 0.15))
            else:
                dims = (int(rng.integers(5, 80)), int(rng.integers(6, 120)), 8 * int(rng.integers(1, 30)))
            dtype = "bf16" if bf16 else "f64"
            tdt = torch.bfloat16 if bf16 else torch.float64
            w = L.effective_weights(shape)
            w = w / w.sum()
            a = torch.from_numpy(rng.standard_normal(L.padded_shape(shape, dims))).to(tdt).cuda()
            times = int(pick([2, 3, 6, 7, 12, 13, 25]))
            ref, _ = run(shape, dims, dtype, w, a, times, {"torus": 0}, "periodic")
            got, sig = run(shape, dims, dtype, w, a, times, {"torus": 1}, "periodic")
            ok, what = close(got, ref, 0 if bf16 else 1e-12), (shape, dims, dtype, times, "periodic", sig)
        else:  # general49
            dims = (int(rng.integers(1, 600)), 2 * int(rng.integers(1, 500)) + int(rng.random() < 0.15))
            w = rng.standard_normal(49)
            w /= np.abs(w).sum()
            a = torch.from_numpy(rng.standard_normal(L.padded_shape("box2d3r", dims))).cuda()
            times = int(pick([6, 7, 10, 12, 17, 23]))
            ref, _ = run("box2d3r", dims, "f64", w, a, times, {"steps_per_launch": 1})
            got, sig = run("box2d3r", dims, "f64", w, a, times, {})
            ok, what = close(got, ref, 1e-12), ("box2d3r 49 taps", dims, times, sig)
    except L.LoraError as e:
        if "status -2" in str(e) or "unsupported" in str(e).lower():
            continue
        print("ERROR", kind, str(e)[:200], flush=True)
        bad += 1
        continue
    n += 1
    kinds[kind] = kinds.get(kind, 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH", kind, what, flush=True)
    if n % 40 == 0:
        print(f"... {n} cases, {bad} bad, {kinds}", flush=True)
print(f"fuzz_r04: {n - bad}/{n} cases agree {kinds}")
sys.exit(1 if bad else 0)
