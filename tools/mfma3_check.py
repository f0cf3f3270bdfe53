#!/usr/bin/env python3
"""Development tool: the bf16 matrix-pipe variant of box3d1r (LORA_VARIANT_MFMA) against its oracle contract, then timed
against the vector kernel on 768^3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run(shape, bits, times, weights, variant, opts=None):
    h = L.ops.halo(shape)
    dims = tuple(bits.shape[i] - 2 * h[i] for i in range(3))
    plan = L.Plan(shape, dims, dtype="bf16").set_weights(weights)
    plan.set_variant(variant)
    for k, v in (opts or {}).items():
        plan.set_option(k, v)
    b0 = torch.from_numpy(bits.view(np.int16)).cuda()
    b1 = torch.zeros_like(b0)
    plan.run(b0, b1, times)
    torch.cuda.synchronize()
    return (b0, b1)[times % 2].cpu().numpy().view(np.uint16), plan.kernel_name


bad = 0
rng = np.random.default_rng(3)
shape = "box3d1r"
w0 = O.effective_weights(shape)
for dims in ((8, 28, 64), (9, 31, 248), (5, 3, 8), (37, 64, 360), (40, 61, 128), (70, 90, 72)):
    for wname, w in (("ref", w0), ("norm", w0 / w0.sum())):
        for dname in ("int", "gauss"):
            ps = O.padded_shape(shape, dims)
            a = rng.integers(0, 100, ps).astype(np.float64) if dname == "int" else rng.standard_normal(ps)
            bits = O.to_bf16(a)
            for times in (2, 4, 5):
                if wname == "ref" and times > 4:
                    continue
                got, kname = run(shape, bits, times, w, L.VARIANT_MFMA)
                exp = O.run_bf16(shape, bits, times, weights=w, separable="mfma")
                same = np.array_equal(got, exp)
                g, e = O.from_bf16(got), O.from_bf16(exp)
                scale = np.abs(e).max()
                err = np.abs(g - e).max() / (scale if scale > 0 else 1)
                nd = int((got != exp).sum())
                # exact regime (contract bit for bit): integer data through ONE fused launch pair (times = 4 with the
                # normalised taps keeps values small); otherwise within one bf16 ulp of the contract
                exact = dname == "int" and wname == "norm" and times <= 4
                ok = same if exact else err < 2.0 ** -7
                bad += not ok
                print(("ok  " if ok else "FAIL"), kname, dims, wname, dname, times, "identical" if same else f"differs at {nd} of {got.size} (rel {err:.2e})", flush=True)
print("failures:", bad, flush=True)
if bad:
    sys.exit(1)

# timing
dims = (768, 768, 768)
w = w0 / w0.sum()
ps = L.padded_shape(shape, dims)
src = torch.randint(0, 100, ps, device="cuda").to(torch.bfloat16)
dst = torch.zeros_like(src)
pts = 768 ** 3
for name, variant, opts in (("valu", L.VARIANT_DIRECT, {}), ("mfma", L.VARIANT_MFMA, {}), ("mfma zc32", L.VARIANT_MFMA, {"fused_z_chunk": 32}),
                            ("mfma zc128", L.VARIANT_MFMA, {"fused_z_chunk": 128}), ("valu", L.VARIANT_DIRECT, {})):
    plan = L.Plan(shape, dims, dtype="bf16").set_weights(w)
    plan.set_variant(variant)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.step2(src, dst)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        plan.step2(src, dst)
        plan.step2(dst, src)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 1e3 / 10
    print(f'{{"case": "{name}", "kernel": "{plan.kernel_name}", "us": {t * 1e6:.1f}, "gstencils": {2 * pts / t / 1e9:.1f}}}', flush=True)
