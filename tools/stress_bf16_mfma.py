"""Stress of the bf16 matrix-pipe variant against its oracle contract (768 runs): the tool that exposed the vmcnt ordering
assumption recorded in kernels_3d_bf16_mfma.hip (0 failures with the loads-only wait, ~1 in 400 before)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lorastencil_amd as L
from oracle import oracle as O
shape = "box3d1r"
w0 = O.effective_weights(shape)
nbad = 0; nrun = 0
exp_cache = {}
for rep in range(24):
    rng = np.random.default_rng(3)
    for dims in ((8, 28, 64), (9, 31, 248), (5, 3, 8), (37, 64, 360)):
        for wname, w in (("ref", w0), ("norm", w0 / w0.sum())):
            for dname in ("int", "gauss"):
                ps = O.padded_shape(shape, dims)
                a = rng.integers(0, 100, ps).astype(np.float64) if dname == "int" else rng.standard_normal(ps)
                bits = O.to_bf16(a)
                plan = L.Plan(shape, dims, dtype="bf16").set_weights(w); plan.set_variant(L.VARIANT_MFMA)
                for times in (4, 5, 9):
                    if wname == "ref" and times > 4: continue
                    b0 = torch.from_numpy(bits.view(np.int16)).cuda(); b1 = torch.zeros_like(b0)
                    plan.run(b0, b1, times); torch.cuda.synchronize()
                    got = (b0, b1)[times % 2].cpu().numpy().view(np.uint16)
                    key = (dims, wname, dname, times)
                    if key not in exp_cache: exp_cache[key] = O.run_bf16(shape, bits, times, weights=w, separable="mfma")
                    exp = exp_cache[key]
                    g, e = O.from_bf16(got), O.from_bf16(exp)
                    d = np.abs(g - e); nrun += 1
                    if d.max() > 2.0 ** -7 * np.abs(e).max():
                        nbad += 1
                        zz, yy, xx = np.where(d > 2.0 ** -7 * np.abs(e).max())
                        print(rep, dims, wname, dname, times, "max err", d.max(), "count", len(zz), "z", np.unique(zz)[:12], flush=True)
print("bad runs:", nbad, "of", nrun)
