#!/bin/bash
# bench.py lines of every BASELINE config on one GPU (through gpurun, from the repo root) -> gpurun_out/bench_all.jsonl
out=gpurun_out/bench_all.jsonl
: > $out
run() { timeout -k 10 400 python bench.py --cpu-seconds 6 "$@" 2>/dev/null | grep '^{' >> $out; }
run --shape 1d1r --steps 100
run --steps 20 --warmup 5
run --steps 100
run --shape box2d3r --steps 200
run --shape star2d3r --steps 100
run --shape star3d1r --steps 50
run --shape box3d1r --steps 50
run --shape box3d1r --dtype bf16 --steps 50
run --shape box3d1r --dtype bf16 --steps 50 --variant mfma
LORA_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' >> $out
python3 - <<'PY'
import json
for l in open("gpurun_out/bench_all.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["config"]["workload"][:44].ljust(44), d["n_gpus"], d["dtype"], "value", d["value"], "kernel", r["kernel"][:60], "bound", r["bound"], "frac", r["frac"], "hbm", r["hbm"]["frac"], "launch_us", r["launch_us"])
PY
