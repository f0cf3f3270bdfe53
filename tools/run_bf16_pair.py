#!/usr/bin/env python3
"""Profiling driver: the bf16 box3d1r 768^3 fused launch on the vector pipe and on the matrix pipe, back to back."""
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
for name, variant in (("valu", L.VARIANT_DIRECT), ("mfma", L.VARIANT_MFMA)):
    plan = L.Plan(shape, dims, dtype="bf16").set_weights(w)
    plan.set_variant(variant)
    plan.step2(src, dst)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        plan.step2(src, dst)
        plan.step2(dst, src)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 1e3 / 8
    print(f'{{"case": "{name}", "kernel": "{plan.kernel_name}", "us": {t * 1e6:.1f}, "gstencils": {2 * 768 ** 3 / t / 1e9:.1f}}}', flush=True)
