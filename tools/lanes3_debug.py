import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lorastencil_amd as L
from oracle import oracle as O
def where(got, ref):
    bad = np.argwhere(~np.isclose(got, ref, rtol=1e-12, atol=1e-12))
    if len(bad) == 0: return "none"
    return f"{len(bad)} cells, z {bad[:,0].min()}..{bad[:,0].max()} y {bad[:,1].min()}..{bad[:,1].max()} x {bad[:,2].min()}..{bad[:,2].max()} (padded idx); first {bad[0]} got {got[tuple(bad[0])]} ref {ref[tuple(bad[0])]}"
for shape, dims, opts, times in (("star3d1r", (8,16,128), {}, 4), ("star3d1r", (8,16,128), {"fused_z_chunk":5}, 0), ("box3d1r", (8,16,128), {"fused_z_chunk":5}, 0),
                                 ("star3d1r", (41,50,250), {}, 0), ("box3d1r", (41,50,250), {"fused_z_chunk": 64}, 0), ("star3d1r", (16,16,250), {}, 0), ("star3d1r", (16,50,120), {}, 0)):
    a = O.reference_input(shape, dims)
    plan = L.Plan(shape, dims).set_option("steps_per_launch", 4)
    for k, v in opts.items(): plan.set_option(k, v)
    if times:
        b0 = torch.from_numpy(a).cuda(); b1 = torch.zeros_like(b0)
        plan.run(b0, b1, times); torch.cuda.synchronize()
        got = (b0, b1)[times % 2].cpu().numpy(); ref = O.run(shape, a, times)
    else:
        src = torch.from_numpy(a).cuda(); dst = torch.from_numpy(a).cuda(); dst[1:-1,2:-2,4:-4] = -7.0
        plan.stepk(src, dst); torch.cuda.synchronize()
        got = dst.cpu().numpy(); ref = O.run(shape, a, 4)
    print(shape, dims, opts, times, "->", where(got, ref), flush=True)
print("---- K=2 directly")
for shape, dims in (("star3d1r", (8,16,128)), ("box3d1r", (8,16,128)), ("star3d1r", (20,40,250))):
    a = O.reference_input(shape, dims)
    plan = L.Plan(shape, dims).set_option("steps_per_launch", 4)
    src = torch.from_numpy(a).cuda(); dst = torch.from_numpy(a).cuda(); dst[1:-1,2:-2,4:-4] = -7.0
    plan.step2(src, dst); torch.cuda.synchronize()
    got = dst.cpu().numpy(); ref = O.run(shape, a, 2)
    print(shape, dims, "step2 from a ->", where(got, ref), flush=True)
    # second application of step2 on the result
    dst2 = torch.from_numpy(a).cuda(); dst2[1:-1,2:-2,4:-4] = -7.0
    plan.step2(dst, dst2); torch.cuda.synchronize()
    print(shape, dims, "step2 twice ->", where(dst2.cpu().numpy(), O.run(shape, a, 4)), flush=True)
