#!/usr/bin/env python3
"""Timing of the bf16 box3d1r kernels on 768^3: vector kernel vs matrix-pipe variant (development tool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lorastencil_amd as L  # noqa: E402

shape, dims = "box3d1r", (768, 768, 768)
w0 = L.effective_weights(shape)
w = w0 / w0.sum()
ps = L.padded_shape(shape, dims)
src = torch.randint(0, 100, ps, device="cuda").to(torch.bfloat16)
dst = torch.zeros_like(src)
pts = 768 ** 3
cases = [("warm", L.VARIANT_DIRECT, {}), ("valu", L.VARIANT_DIRECT, {}), ("mfma", L.VARIANT_MFMA, {}),
         ("mfma nosplit", L.VARIANT_MFMA, {"mfma_split": 0}), ("mfma zc32", L.VARIANT_MFMA, {"fused_z_chunk": 32}),
         ("mfma zc20", L.VARIANT_MFMA, {"fused_z_chunk": 20}), ("valu", L.VARIANT_DIRECT, {}), ("mfma", L.VARIANT_MFMA, {})]
for extra in sys.argv[1:]:
    k, v = extra.split("=")
    cases.append((f"mfma {extra}", L.VARIANT_MFMA, {k: int(v)}))
for name, variant, opts in cases:
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
