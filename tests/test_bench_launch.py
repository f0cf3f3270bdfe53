"""bench.py's launch path: `python bench.py --gpus N` must start its own N ranks (the driver's N = 1 form of the command,
with N > 1) and print ONE JSON line from rank 0.  On a box without a GPU only the plumbing can run
(`--plumbing-only`: rendezvous + one reduction, value null -- the engine has no CPU path to measure); the same command
with real sweeps is the GPU test below."""
import json
import os
import subprocess
import sys

import pytest
from conftest import ROOT


def _bench(*args, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e,
                       timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_starts_its_own_ranks():
    r, lines = _bench("--gpus", "2", "--steps", "4", "--warmup", "0", "--plumbing-only")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["plumbing_only"] is True and d["value"] is None


def test_planned_launches_follow_the_driver_rule():
    sys.path.insert(0, ROOT)
    import bench

    # (K-application launches, two-application launches, single sweeps) of a slab run: SlabDriver.run's rule
    assert bench.planned_launches(1, 100, 8) == (12, 0, 4)   # 1D: 12 eight-step launches + 4 single sweeps
    assert bench.planned_launches(2, 100, 2) == (50, 0, 0)
    assert bench.planned_launches(2, 21, 2) == (10, 0, 1)
    assert bench.planned_launches(2, 20, 4) == (5, 0, 0)      # an odd number of fused launches is fine in a slab run
    assert bench.planned_launches(2, 23, 4) == (5, 1, 1)      # 2D, K = 4: a two-application launch for the tail
    assert bench.planned_launches(2, 23, 6) == (3, 1, 1)      # K = 6: 18 + a four-application tail + one single sweep
    assert bench.planned_launches(2, 21, 6) == (3, 1, 1)      # ... 18 + a two-application tail + one
    assert bench.planned_launches(3, 23, 4) == (5, 1, 1)      # 3D, K = 4: tails of two
    from lorastencil_amd.slab import launch_depths
    assert launch_depths(2, 6, 23) == [6, 6, 6, 4, 1]
    assert launch_depths(2, 6, 23, tails=(2,)) == [6, 6, 6, 2, 2, 1]   # a stepper without four-application launches
    assert launch_depths(2, 6, 9, steps_done=1) == [1, 6, 2]            # a resumed run starts fusing at an even level
    assert launch_depths(2, 6, 9, fused=False) == [1] * 9
    assert bench.planned_launches(3, 7, 2) == (3, 0, 1)
    assert bench.planned_launches(2, 20, 1) == (0, 0, 20)


@pytest.mark.gpu
def test_bench_two_ranks_over_gloo_on_one_gpu():
    r, lines = _bench("--gpus", "2", "--steps", "4", "--warmup", "0", "--size", "256", "256", "--no-cpu-baseline",
                      env={"LORA_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
def test_bench_line_roofline_is_a_fraction():
    r, lines = _bench("--steps", "8", "--warmup", "2", "--size", "2048", "2048", "--cpu-seconds", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[-1])
    rf = d["roofline"]
    # 8 sweeps = 6 + 2, which the driver runs as 4 + 4: no full-depth launch, two four-application ones
    assert d["n_gpus"] == 1 and 0 < rf["frac"] <= 1.0 and rf["applications_per_launch"] == 4
    assert 0 < rf["hbm"]["frac"] <= 1.0 and rf["bound"] in ("hbm", "fp64_valu") and rf["limiter"] in ("hbm", "valu", "lds")
    # the arithmetic roof is priced on EXECUTED instructions (the kernel's own PMC profile) or not at all: the direct-form
    # flops the low-rank evaluation never performs are reported, labelled, and never become `frac` beyond what the pipe can do
    fp = rf["fp64"]
    assert fp["useful_direct_form_tflops"] > 0 and fp["flops_per_point_direct_form"] == 49
    if fp["executed"] is not None:
        assert 0 < fp["executed"]["frac"] <= 1.0 and fp["executed"]["ops_per_point"] < 49
    if rf["bound"] == "hbm":
        assert rf["frac"] == rf["hbm"]["frac"] and rf["unit"] == "GB/s"
    else:
        assert rf["unit"] == "TFLOP/s" and rf["frac"] <= 1.0
    assert abs(rf["frac_one_sweep_equiv"] - 4 * rf["hbm"]["frac"]) < 1e-3 and rf["launches"] == 2
    assert rf["tail_fused_launches"] == 2 and rf["clock_ramp_note"] is not None
    assert rf["traffic"] is None or rf["traffic_key"].endswith(rf["kernel"])  # never a number of another kernel
    assert d["cpu_baseline"]["cores"] >= 1
