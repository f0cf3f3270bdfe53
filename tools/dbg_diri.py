import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lorastencil_amd as L
rng = np.random.default_rng(3)
for shape, dims in (("star2d1r", (64, 128)),):
    w = L.effective_weights(shape); w = w / w.sum()
    a = rng.standard_normal(L.padded_shape(shape, dims))
    outs = []
    for stream in (0, 1):
        plan = L.Plan(shape, dims).set_weights(w).set_boundary("dirichlet")
        plan.set_option("stream", stream)
        src = torch.from_numpy(a).cuda(); dst = torch.from_numpy(a).cuda()
        plan.step2(src, dst); torch.cuda.synchronize()
        outs.append(dst.cpu().numpy())
    d = np.abs(outs[0] - outs[1]) > 1e-12
    print("bad", d.sum())
    rows = np.where(d.any(axis=1))[0]; cols = np.where(d.any(axis=0))[0]
    print("rows", rows.tolist()); print("cols", cols.tolist()[:40], "...", cols.tolist()[-10:])
