import sys, time
sys.path.insert(0, '/root/repo')
import torch, lorastencil_amd as L
shape, dims = "star2d1r", (16384, 16384)
w = L.effective_weights(shape); w = w / w.sum()
ps = L.padded_shape(shape, dims)
src = torch.rand(ps, dtype=torch.float64, device="cuda"); dst = torch.zeros_like(src)
plan = L.Plan(shape, dims).set_weights(w)
# warm the GPU with another kernel family
p1 = L.Plan(shape, dims).set_weights(w).set_option("steps_per_launch", 1)
for _ in range(5): p1.step(src, dst)
torch.cuda.synchronize()
for K in (4, 6, 2):
    plan.set_option("steps_per_launch", 6)
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        if K == 6: plan.stepk(src, dst)
        else:
            plan.set_option("wg", 1).set_option("steps_per_launch", K); plan.stepk(src, dst)
        t1 = time.perf_counter()
        e1.record(); torch.cuda.synchronize()
        print(f"K={K} launch {i}: host {1e6*(t1-t0):.0f} us, device {1e3*e0.elapsed_time(e1):.0f} us", flush=True)
    plan.set_option("wg", -1)
