#!/usr/bin/env python3
"""Headline benchmark: GStencils/s of the stencil sweep, star2d1r 16384 x 16384 fp64 (BASELINE.json configs[1]),
on N MI355X GPUs of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one kernel application (one sweep of the whole grid): the reference's time-step loop body
(2d/gpu.cu:544-546).  `value` counts interior points x applications per second with F = 1 (the reference's own
printout multiplies by 3 for this shape, 2d/gpu.cu:553; that figure is reported as `value_reference_convention`).
Inputs are synthetic small integers (0..99, the range of the reference's rand()%100 fill) generated on the device
and resident in HBM before the timed region.  For N > 1 the SAME 16384^2 grid is cut into N row slabs ("strong"
scaling, as BASELINE.json quotes the metric); ghost zones several launches deep are refreshed by one RCCL exchange every
E launches, hidden behind the next launch's interior (lorastencil_amd/slab.py).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (a
torch.distributed.run child, before this process touches the GPU) and relays rank 0's line.

The single JSON line also carries
  roofline      for the dominant kernel, against both roofs (`hbm`, `fp64`; `bound` / `frac` name the nearer one).  hbm:
                COMPULSORY bytes per launch (one read + one write of every interior point, 2 x sizeof(T) x points,
                however many applications the launch fuses) / its average launch duration (HIP events on the launch
                stream around the fused launches of the timed region, lora_plan_run_profiled), against the 8 TB/s
                HBM3E peak: <= 1 by construction.  fp64: algorithmic flops (direct form) x applications per
                launch / the same duration, against the 78.6 TFLOP/s fp64 FMA peak (matrix = vector rate on gfx950).
                `frac_one_sweep_equiv` is SURVEY 8d's figure (16 B per point per APPLICATION: > 1 means the fused
                launch beats what any one-sweep-per-launch kernel could reach); `copy_bw_frac` relates `achieved` to
                a torch copy_ of the same grid timed in this run; `traffic` = measured HBM bytes per launch (rocprofv3
                PMC passes, profiles/pmc_traffic.json, keyed by the kernel's full signature: null when that exact
                instantiation has not been measured);
                `valu_busy` / `mfma_busy` / `waves_per_simd` = pipe utilisation from the SQ passes of the same profile.
                Launch-bound grids (<= 64 MB, >= 16 sweeps: the 1D configuration) are timed as the product runs them
                by default -- one hipGraph replay of the K sweeps, captured before the timed region
                (`config.graph_replay`); their per-launch durations come from an untimed profiled run of the same K;
  cpu_baseline  the CPU oracle (a port of the reference's test_cpu loop) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec; ~6.3 TB/s achievable copy)
FP64_PEAK_TFLOPS = 78.6  # fp64 FMA peak: on gfx950 the matrix (MFMA) rate equals the vector rate (256 CUs x 4 SIMDs x 16
#                          lanes x 2 flop x 2.4 GHz; DESIGN.md section 4 measures 64 cycles per v_mfma_f64_16x16x4)
# algorithmic flops per point per application, direct form 2 nnz - 1
FLOPS_PER_POINT = {"1d1r": 13, "1d2r": 17, "star2d1r": 49, "box2d3r": 97, "box2d1r": 97, "star2d3r": 25, "star3d1r": 13,
                   "box3d1r": 53}
OVERFLOW_STEPS = {"star2d1r": 140, "box2d3r": 120, "box2d1r": 120, "star2d3r": 200, "star3d1r": 300, "box3d1r": 180,
                  "1d1r": 240, "1d2r": 200}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="star2d1r")
    ap.add_argument("--size", type=int, nargs="+", default=None, help="interior sizes (default: the BASELINE config)")
    ap.add_argument("--option", action="append", default=[], help="plan option key=value (e.g. rows_per_thread=4)")
    ap.add_argument("--variant", choices=["auto", "direct", "mfma"], default="auto")
    ap.add_argument("--dtype", choices=["f64", "bf16"], default="f64", help="bf16: 3D shapes only (BASELINE config 5)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="start the ranks, rendezvous, reduce a clock and print a line with value null: checks the "
                         "launch path on a box without a GPU (no kernel runs, nothing is measured)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


DEFAULT_SIZES = {"star2d1r": (16384, 16384), "box2d3r": (8192, 8192), "box2d1r": (8192, 8192), "star2d3r": (16384, 16384),
                 "star3d1r": (512, 512, 512), "box3d1r": (768, 768, 768), "1d1r": (1048576,), "1d2r": (1048576,)}


def cpu_baseline(shape, dims, budget_s):
    """Oracle (CPU restatement of the reference's naive loop) on a bounded sample: the first rows/planes of the same
    grid, all host threads, at least one full sweep of the sample."""
    import numpy as np

    from oracle import oracle as O

    threads = O.max_threads()
    w = O.effective_weights(shape)
    inner = 1
    for d in dims[1:]:
        inner *= d
    rng = np.random.default_rng(0)
    # calibrate on a thin slab, then size the sample so that one sweep takes about a quarter of the budget
    n_cal = int(max(8, min(dims[0], (4 << 20) // max(inner, 1))))
    cal = rng.integers(0, 100, O.padded_shape(shape, (n_cal,) + tuple(dims[1:]))).astype(np.float64)
    cal_out = np.zeros_like(cal)
    O.step(shape, cal, w, out=cal_out, threads=threads)
    tc = time.perf_counter()
    O.step(shape, cal, w, out=cal_out, threads=threads)
    rate = n_cal * inner / max(time.perf_counter() - tc, 1e-6)  # points per second
    n0 = int(max(n_cal, min(dims[0], rate * budget_s / 4.0 // max(inner, 1))))
    sdims = (n0,) + tuple(dims[1:])
    a = rng.integers(0, 100, O.padded_shape(shape, sdims)).astype(np.float64)
    out = np.zeros_like(a)
    O.step(shape, a, w, out=out, threads=threads)  # warm-up (page faults, thread pool)
    reps, t0 = 0, time.perf_counter()
    while True:
        O.step(shape, a, w, out=out, threads=threads)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s / 2 or reps >= 50:
            break
    pts = 1
    for d in sdims:
        pts *= d
    val = pts * reps / el / 1e9
    # single-thread figure on a smaller piece (the reference's test_cpu is single-threaded)
    n1 = max(4, n0 // max(2 * threads, 1))
    a1 = a[: n1 + (a.shape[0] - n0)]
    o1 = np.zeros_like(a1)
    t1 = time.perf_counter()
    O.step(shape, np.ascontiguousarray(a1), w, out=o1, threads=1)
    el1 = time.perf_counter() - t1
    pts1 = n1 * inner
    # the reference's OWN checker (oracle/_ref, built from its sources where they are available), same piece, 1 thread
    ref_rate = None
    try:
        from oracle import ref as R

        if R.available():
            a1c = np.ascontiguousarray(a1)
            t2 = time.perf_counter()
            R.test_cpu(a1c, O.default_params(shape), o1)
            ref_rate = round(pts1 / (time.perf_counter() - t2) / 1e9, 4)
    except Exception:  # the _ref build is optional on the GPU box
        ref_rate = None
    return {
        "value": round(val, 4),
        "unit": "GStencils/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{shape} sub-grid {'x'.join(str(d) for d in sdims)} of the {'x'.join(str(d) for d in dims)} "
                  f"workload, {reps} sweeps, OpenMP over rows, same taps/order as the reference's test_cpu",
        "value_1thread": round(pts1 / el1 / 1e9, 4),
        "reference_test_cpu_1thread": ref_rate,  # kind "reference": the reference's test_cpu itself, single-threaded
        "host_cpus": os.cpu_count(),
    }


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed launch: start the N ranks ourselves, as a CHILD
    process, before this one has made any GPU call (a process that has initialised the GPU must not exec another
    program on this pool), and pass rank 0's JSON line and the children's exit status through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if (proc.returncode != 0 or line is not None) else 1


def planned_launches(ndim, times, spl, fused_ok=True):
    """(K-application launches, tail launches, single sweeps) of one SLAB run (world > 1), counted from the drivers' own
    rule (lorastencil_amd.slab.launch_depths, which SlabDriver.run and BlockDriver.run execute and slab.cpp / blocks.cpp
    restate).  (The single-GPU schedule, with its scratch-grid routing, is reported by lora_plan_run_profiled itself.)"""
    from lorastencil_amd.slab import launch_depths
    depths = launch_depths(ndim, spl, times, fused=fused_ok and spl >= 2)
    nk = sum(1 for d in depths if d == spl and spl > 1)
    ns = sum(1 for d in depths if d == 1) if spl > 1 else len(depths)
    return nk, len(depths) - nk - ns, ns


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the engine normally arrives prebuilt with the snapshot; if it did not, local rank 0 builds it (hipcc), the
    # other ranks wait for the library to appear -- there is no other implementation to fall back to
    lib_path = os.path.join(ROOT, "lorastencil_amd", "lib", "liblorastencil_hip.so")
    if not os.path.exists(lib_path):
        if local_rank == 0:
            import __graft_entry__ as g

            g.build(only_if_missing=True)
        else:
            t_wait = time.time()
            while not os.path.exists(lib_path) and time.time() - t_wait < 900:
                time.sleep(2)
            time.sleep(5)  # let the linker finish writing

    args.gpus = world  # inside a launch the launcher's world size is authoritative
    if args.plumbing_only:
        # the N-rank launch path without the engine: rendezvous over gloo, one MAX-reduction, one line from rank 0
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        t = torch.tensor([time.perf_counter()], dtype=torch.float64)
        if world > 1:
            dist.init_process_group("gloo")
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "GStencils/s", "value": None, "unit": "GStencils/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "plumbing_only": True}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    import lorastencil_amd as L
    from lorastencil_amd import slab

    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    # LORA_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("LORA_DIST_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    shape = args.shape
    if args.dtype == "bf16" and shape == "star2d1r":
        shape = "box3d1r"  # the bf16 configuration of BASELINE.json
    dims = tuple(args.size) if args.size else DEFAULT_SIZES[shape]
    bf16 = args.dtype == "bf16"
    tdtype = torch.bfloat16 if bf16 else torch.float64
    esize = 2 if bf16 else 8
    K, W = args.steps, args.warmup
    opts = dict(kv.split("=") for kv in args.option)
    variant = {"auto": L.VARIANT_AUTO, "direct": L.VARIANT_DIRECT, "mfma": L.VARIANT_MFMA}[args.variant]

    # taps: the operator's own; if the run is longer than the fp64 range allows use the normalised taps
    weights = L.effective_weights(shape)
    normalised = (K + W) > (20 if bf16 else OVERFLOW_STEPS.get(shape, 100))
    if normalised:
        weights = weights / weights.sum()

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1:
        plan = L.Plan(shape, dims, dtype=args.dtype).set_weights(weights)
        if variant != L.VARIANT_AUTO:
            plan.set_variant(variant)
        for k, v in opts.items():
            plan.set_option(k, int(v))
        ps = plan.padded_shape
        src0 = torch.randint(0, 100, ps, generator=gen, device=dev).to(tdtype)
        b0, b1 = src0.clone(), torch.zeros(ps, dtype=tdtype, device=dev)
        run = lambda n: plan.run(b0, b1, n)  # noqa: E731
        kernel = plan.kernel_name
        signature = plan.kernel_signature
        local_points = 1
        for d in dims:
            local_points *= d

        def reset():
            b0.copy_(src0)
            b1.zero_()
    else:
        # options and variant go in BEFORE the driver fixes its ghost depth (they decide applications per launch)
        drv = slab.SlabDriver(shape, dims, device=dev, weights=weights, dtype=args.dtype, options=opts,
                              variant=None if variant == L.VARIANT_AUTO else variant)
        plan = drv.stepper.plan
        assert drv.stepper.apps_per_launch == (drv.apps if drv.fused else drv.stepper.apps_per_launch)
        src0 = torch.randint(0, 100, drv.local_padded_shape, generator=gen, device=dev).to(tdtype)
        run = lambda n: drv.run(n)  # noqa: E731
        kernel = plan.kernel_name
        signature = plan.kernel_signature
        local_points = drv.layout.own
        for d in drv.layout.local_dims[1:]:
            local_points *= d

        def reset():
            drv.load_local(src0)
            drv.refresh_ghosts()  # neighbours' own rows of time level 0 into the ghost zones

    # achievable-bandwidth reference of THIS box, in this run: a device copy of the same grid (read + write once)
    copy_gbs = None
    if world == 1:
        dst_c = torch.empty_like(src0)
        dst_c.copy_(src0)
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            dst_c.copy_(src0)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * src0.numel() * esize * 5 / (c0.elapsed_time(c1) / 1e3) / 1e9
        del dst_c

    if world == 1:
        # set-up, not sweeps: what the K-sweep run allocates on first need (the scratch grid of an odd number of fused
        # launches: a hipMalloc + hipMemset of one more grid, which on a fresh device has taken 60 ms) exists before the
        # warm-up, so that nothing but the reset of the buffers lies between the warm-up and the timed sweeps
        plan.prepare_run(K)
        torch.cuda.synchronize()
    reset()
    run(W)
    reset()  # keep the value range of the timed steps independent of the warm-up length
    # Launch-bound grids (<= 64 MB padded, >= 16 sweeps: the 1D configuration) run as the product runs them by default:
    # lora_plan_run captures its launches into a hipGraph on first use and replays it.  The capture is set-up work, done
    # once before the timed region by an untimed run of the same K sweeps on a side stream (graphs need a real stream);
    # the timed region is then one replay = exactly K sweeps.  The per-launch durations for the roofline come from a
    # separate, untimed profiled run (events between launches cannot be recorded inside a graph replay).
    graph_replay = False
    side = None
    if world == 1 and K >= 16 and src0.numel() * esize <= (64 << 20) and plan.get_option("graph") != 0:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            plan.run(b0, b1, K, stream=side)
        side.synchronize()
        reset()
        torch.cuda.synchronize()
        graph_replay = True
    def timed_region():
        """EXACTLY K sweeps between barrier + synchronize on both sides; the MAX over ranks."""
        barrier()
        prof_ = None
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        if graph_replay:
            plan.run(b0, b1, K, stream=side)
            side.synchronize()
        elif world == 1:
            prof_ = plan.run_profiled(b0, b1, K)  # = plan.run + events around its fused / single-sweep launches; blocks
        else:
            run(K)
        ev1.record()
        barrier()
        t1 = time.perf_counter()
        el = t1 - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ev0.elapsed_time(ev1), prof_

    elapsed, ev_ms, prof = timed_region()
    # Throttle guard.  Twice in some fifty runs of this round a whole timed region ran FOUR times slower than every run
    # around it on the same code (83 ms instead of 21 for the 100 sweeps; the copy-bandwidth reference measured a moment
    # earlier in the same process was normal): the box, not the kernels.  The yardstick is this run's own: ONE more fused
    # launch right behind the timed region (untimed for the metric).  If the timed sweeps ran more than twice as slow per
    # sweep as that launch, the region is timed ONCE more and both figures are reported.
    retimed = None
    probe_ms_per_sweep = None
    if not graph_replay:
        n_probe = max(1, int(plan.get_option("steps_per_launch")))
        n_probe += n_probe & 1  # (an even count: the buffers' halo state as at the start of a run)
        reset()
        p_ev0, p_ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        p_ev0.record()
        run(n_probe)
        p_ev1.record()
        barrier()
        probe_ms_per_sweep = p_ev0.elapsed_time(p_ev1) / n_probe
    slow = bool(probe_ms_per_sweep and elapsed * 1e3 / K > 2.0 * probe_ms_per_sweep)
    if world > 1:
        f = torch.tensor([1.0 if slow else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        slow = bool(f.item() > 0)
    if slow:
        first_ms = elapsed * 1e3
        reset()
        elapsed, ev_ms, prof = timed_region()
        retimed = {"first_region_ms": round(first_ms, 3), "probe_ms_per_sweep": round(probe_ms_per_sweep, 4),
                   "reason": "the first timed region ran > 2 x slower per sweep than one more launch right behind it: timed once more"}
    if graph_replay:
        reset()
        torch.cuda.synchronize()
        prof = plan.run_profiled(b0, b1, K)  # untimed: launch counts and durations of the same schedule

    points = 1
    for d in dims:
        points *= d
    value = points * K / elapsed / 1e9
    # The dominant kernel and its average launch duration over the timed region.  One launch applies `spl` sweeps
    # (temporal fusion) but moves the grid through HBM once: compulsory bytes per launch = 2 x sizeof(T) x points.
    if world == 1:
        spl = plan.get_option("steps_per_launch")
        if prof.fused_launches > 0:
            launches, apps, launch_s = prof.fused_launches, prof.apps_per_fused_launch, prof.fused_ms / 1e3 / prof.fused_launches
            if apps != spl:  # 1D: lora_plan_run fuses 16 / 32 sweeps per launch on long runs (the plan's own depth is 8)
                signature = signature.replace(f"k={spl}", f"k={apps}")
                spl = apps
        elif prof.two_launches > 0:
            # a run too short for a full-depth launch: its shallower fused launches (2D: four / two applications) are the
            # dominant kernel; their average depth is what the run's sweeps leave after the single sweeps
            launches = prof.two_launches
            apps = max((K - prof.single_launches) // prof.two_launches, 1)
            launch_s = prof.two_ms / 1e3 / prof.two_launches
            signature = signature.replace(f"k={spl}", f"k={apps}")
            spl = apps
        else:
            launches, apps, launch_s = max(prof.single_launches, 1), 1, prof.single_ms / 1e3 / max(prof.single_launches, 1)
            single = L.Plan(shape, dims, dtype=args.dtype).set_weights(weights).set_option("steps_per_launch", 1)
            kernel, signature = single.kernel_name, single.kernel_signature
    else:
        spl = drv.apps if drv.fused else 1
        nf, n2, ns = planned_launches(len(dims), K, spl, drv.fused)
        # slab launches can be issued from Python in pieces (interior first, ends after the deferred wait): count
        # applications, not pieces;
        # the K-application launches' share of the timed region is taken as their share of the sweeps
        launches, apps = (nf, spl) if nf else (max(ns + n2, 1), 1)
        launch_s = ev_ms / 1e3 * (nf * spl / K if nf else 1.0) / max(launches, 1)
    bytes_per_launch = local_points * 2.0 * esize          # compulsory: one read + one write of the grid
    achieved = bytes_per_launch / launch_s / 1e9
    one_sweep_equiv = achieved * apps                       # 2 x sizeof(T) per point per APPLICATION
    # the other roof is arithmetic; everything about it is derived below from the counters of this kernel's profile
    flops_per_launch = float(FLOPS_PER_POINT.get(shape, 0)) * local_points * apps   # direct form: "useful", NOT executed
    useful_tflops = flops_per_launch / launch_s / 1e12
    hbm_frac = achieved / HBM_PEAK_GBS
    traffic, pmc = None, {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    tkey = f"{shape}:{'x'.join(map(str, dims))}:{world}:{signature}"
    if os.path.exists(tpath):
        try:
            entry = json.load(open(tpath)).get(tkey)  # a miss stays null: never a number measured on another kernel
        except Exception:
            entry = None
        # an entry is either the HBM byte count alone (older profiles) or an object that also carries the pipe
        # utilisation from the SQ counter passes of the same profile (tools/pmc_summary.py)
        if isinstance(entry, dict):
            traffic, pmc = entry.get("hbm_bytes"), entry
        elif entry is not None:
            traffic, pmc = entry, {}

    # ---- what the silicon did (VERDICT r03 #2 / ADVICE r03) ------------------------------------------------------------
    # The kernels evaluate a low-rank FORM of the stencil (nested profiles, pyramid terms, separable passes): far fewer
    # operations than the direct form's 2 nnz - 1 flops per point.  The arithmetic roof is therefore priced on EXECUTED
    # instructions -- SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 per launch from the kernel's own PMC profile (64 lanes x 2 flops per
    # FMA, x 1 per add / multiply), divided by THIS run's launch duration -- against the fp64 peak (spec, 2.4 GHz) and
    # against the peak at the clock the chip held under the kernel.  The direct-form figure stays, labelled `useful`.
    kinds = pmc.get("valu_insts_by_kind") or {}
    p32 = bf16
    sfx = "f32" if p32 else "f64"
    executed = None
    if kinds.get(f"fma_{sfx}") is not None:
        n_fma, n_add, n_mul = kinds.get(f"fma_{sfx}", 0), kinds.get(f"add_{sfx}", 0), kinds.get(f"mul_{sfx}", 0)
        ex_flops = 64.0 * (2.0 * n_fma + n_add + n_mul)
        clock = pmc.get("clock_ghz")
        # fp64: 16 lanes x 2 flops per SIMD and cycle (v_fma_f64 issues over 4 cycles); fp32: the same lanes, two values each
        # when packed (v_pk_fma_f32) -- the 157.3 TFLOP/s figure -- so the fp32 roof is priced on the packed rate
        peak_spec = FP64_PEAK_TFLOPS * (2.0 if p32 else 1.0)
        peak_clk = (clock * 256 * 4 * 16 * 2 * (2.0 if p32 else 1.0) / 1e3) if clock else None
        executed = {
            "tflops": round(ex_flops / launch_s / 1e12, 2),
            "frac": round(ex_flops / launch_s / 1e12 / peak_spec, 4),
            "peak_at_clock_tflops": round(peak_clk, 1) if peak_clk else None,
            "frac_at_clock": round(ex_flops / launch_s / 1e12 / peak_clk, 4) if peak_clk else None,
            "ops_per_point": round(64.0 * (n_fma + n_add + n_mul) / (local_points * apps), 2),
            "insts_per_launch": {"fma": n_fma, "add": n_add, "mul": n_mul},
        }
    traffic_gbs = traffic / launch_s / 1e9 if traffic else None
    # which resource the kernel uses most of what it can get: the vector pipe's busy share, the LDS array's, and the HBM
    # stream (measured bytes when there are any) against a plain copy of the same grid on this box in this process
    util = {"hbm": ((traffic_gbs or achieved) / copy_gbs) if copy_gbs else hbm_frac,
            "valu": pmc.get("valu_busy"), "lds": pmc.get("lds_active")}
    limiter = max((k for k in util if util[k] is not None), key=lambda k: util[k])
    arith_name = "fp32_valu" if p32 else "fp64_valu"
    if executed is not None:
        arith_frac, arith_achieved = executed["frac"], executed["tflops"]
        arith_peak = FP64_PEAK_TFLOPS * (2.0 if p32 else 1.0)
    else:  # no counters for this instantiation: the direct-form figure, capped by what the pipe can do
        arith_frac, arith_achieved, arith_peak = (0.0 if p32 else min(useful_tflops / FP64_PEAK_TFLOPS, 1.0)), useful_tflops, FP64_PEAK_TFLOPS
    if util["valu"] is not None:
        bound_hbm = util["hbm"] >= util["valu"]
    else:
        bound_hbm = hbm_frac >= arith_frac
    if rank == 0:
        res = {
            "metric": "GStencils/s",
            "value": round(value, 3),
            "unit": "GStencils/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed * 1e3 / K, 6),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{shape} {'x'.join(map(str, dims))} {'bf16' if bf16 else 'fp64'}, {K} sweeps (lorastencil_{len(dims)}d {shape} "
                            f"{' '.join(map(str, dims))} {K})",
                "parallelism": (f"row-slabs x{world}, ghost {drv.layout.ghost} rows refreshed every "
                                f"{drv.exchange_every} launches ({drv.exchange_mode}, "
                                f"{'boundary strips first' if drv.overlap else 'whole slab per launch'}, deferred wait)")
                               if world > 1 else "single GPU",
                "kernel": kernel,
                "variant": {1: "direct", 2: "mfma"}.get(plan.get_option("variant"), "?"),
                "normalised_taps": bool(normalised),
                "steps_per_launch": spl,
                "graph_replay": graph_replay,
            },
            "value_reference_convention": round(value * L.ops.gstencil_factor(shape), 3),
            "roofline": {
                # `bound`: the resource this kernel uses the larger share of -- the HBM stream (measured bytes per launch
                # against a plain copy on this box) or the vector pipe (SQ_ACTIVE_INST_VALU busy share) -- from the kernel's
                # PMC profile; without one, the nearer roof.  `achieved` / `peak` / `frac` are that roof's: hbm = COMPULSORY
                # bytes per launch / launch duration against 8 TB/s; fp64_valu = EXECUTED flops per launch / the same duration
                # against the fp64 FMA peak (also the MFMA f64 rate: one shared datapath, profiles/r03_fp64_coissue_probe.txt)
                "bound": "hbm" if bound_hbm else arith_name,
                "limiter": limiter,
                "utilisation": {k: (round(v, 3) if v is not None else None) for k, v in util.items()},
                "kernel": signature,
                "achieved": round(achieved, 1) if bound_hbm else round(arith_achieved, 2),
                "peak": HBM_PEAK_GBS if bound_hbm else arith_peak,
                "unit": "GB/s" if bound_hbm else "TFLOP/s",
                "frac": round(hbm_frac if bound_hbm else arith_frac, 4),
                "hbm": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4)},
                ("fp32" if p32 else "fp64"): {
                    "executed": executed,
                    "useful_direct_form_tflops": round(useful_tflops, 2),
                    "useful_direct_form_frac": round(useful_tflops / FP64_PEAK_TFLOPS, 4) if not p32 else None,
                    "flops_per_point_direct_form": FLOPS_PER_POINT.get(shape),
                    "peak": FP64_PEAK_TFLOPS * (2.0 if p32 else 1.0), "unit": "TFLOP/s",
                    "note": "executed = SQ_INSTS_VALU_{FMA,ADD,MUL} of this kernel's profile x 64 lanes (FMA = 2 flops) / this run's "
                            "launch duration; the direct-form figure counts flops the low-rank evaluation never performs"},
                "traffic": traffic,
                # the same launch duration applied to the MEASURED bytes
                "traffic_gbs": round(traffic_gbs, 1) if traffic_gbs else None,
                "traffic_key": tkey,
                # fraction of SIMD issue cycles spent on vector-ALU / matrix instructions, from the SQ counter passes of
                # the profile `traffic` comes from (null when that profile has none): the utilisation of the pipe the
                # "mfma" roof prices against
                "valu_busy": pmc.get("valu_busy"),
                "mfma_busy": pmc.get("mfma_busy"),
                "waves_per_simd": pmc.get("waves_per_simd"),
                # of the vector pipe's issue slots, the share that issued fp64 arithmetic; and the shader clock the chip
                # held under this kernel (GRBM_GUI_ACTIVE / 8 / duration) -- both from the same profile
                "valu_issue_frac": pmc.get("fp64_issue_frac"),
                "lds_active": pmc.get("lds_active"),
                "wait_inst_frac": pmc.get("wait_inst_frac"),
                "clock_ghz": pmc.get("clock_ghz"),
                "pmc_source": pmc.get("source"),
                # the chip's clocks ramp up over the first ~50 ms of load and settle after ~100 ms: a shorter
                # timed region measures the ramp, neither the boost nor the sustained rate
                "timed_region_ms": round(elapsed * 1e3, 2),
                "clock_ramp_note": ("timed region < 50 ms: clocks still ramping" if elapsed < 0.05 else None),
                "retimed": retimed,
                "probe_ms_per_sweep": (round(probe_ms_per_sweep, 4) if probe_ms_per_sweep else None),
                "launch_us": round(launch_s * 1e6, 2),
                "launches": launches,
                "bytes_per_launch": round(bytes_per_launch),
                "applications_per_launch": apps,
                "frac_one_sweep_equiv": round(one_sweep_equiv / HBM_PEAK_GBS, 4),
                "copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
                "copy_bw_frac": round(achieved / copy_gbs, 4) if copy_gbs else None,
                "tail_fused_launches": (prof.two_launches if prof else 0),
                "single_sweep_launches": (prof.single_launches if prof else None),
            },
        }
        if world == 1 and not args.no_cpu_baseline and not bf16:
            res["cpu_baseline"] = cpu_baseline(shape, dims, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
