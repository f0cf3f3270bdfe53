#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference host code (run in the build container only).

Inputs are the reference harness's own fills (un-seeded glibc rand(): rand()%10000 in 1D, rand()%100 in 2D/3D,
1d/main.cu:105-109, 2d/main.cu:232-236, 3d/main.cu:164-168) and its own params tables; expected outputs come
from the reference's own ``test_cpu`` (compiled from /root/reference by oracle/build_ref.sh into oracle/_ref),
chained with the reference driver's ping-pong (buffer 1 starts at zero, halos never written: 2d/gpu.cu:531-554).
So step 1 is exactly what the reference's CHECK_ERROR block compares against; steps > 1 extend it with the
driver semantics read from the reference host code.

Each fixture holds data only: `input`, `params`, `dims`, and `out_t<k>` = whole padded buffer [k % 2] after k steps.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as o  # noqa: E402  (fill + params tables, themselves checked against libc / ref)
from oracle import ref  # noqa: E402

CASES = {
    "1d1r": (2048,),
    "1d2r": (2048,),
    "star2d1r": (64, 128),
    "box2d1r": (32, 64),
    "star2d3r": (64, 128),
    "box2d3r": (64, 128),
    "star3d1r": (8, 16, 128),
    "box3d1r": (8, 16, 128),
}
STEPS = (1, 2, 3, 4)


def main():
    if not ref.available():
        sys.exit("oracle/_ref is missing: run oracle/build_ref.sh (needs /root/reference)")
    here = os.path.dirname(os.path.abspath(__file__))
    for shape, dims in CASES.items():
        a = o.reference_input(shape, dims)
        params = o.default_params(shape)
        data = {"input": a, "params": params, "dims": np.array(dims, dtype=np.int64)}
        for t in STEPS:
            data[f"out_t{t}"] = ref.run_chain(a, params, t)
        path = os.path.join(here, f"{shape}.npz")
        np.savez_compressed(path, **data)
        print(f"{shape:10s} dims={dims} -> {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
