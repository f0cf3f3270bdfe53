#!/bin/bash
# bench.py lines (50 sweeps: the sustained clock, not a burst) of one 3D config with the launch cut by option spans3:
#   tools/spans_bench_ab.sh "<bench args>" "<spans3 values>" [alternations]
args=${1:---shape box3d1r --dtype bf16}; vals=${2:-0 2}; n=${3:-3}
for i in $(seq 1 $n); do
for o in $vals; do
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline $args --option spans3=$o 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', 'spans3=$o', d['value'], d['roofline']['launch_us'])"
done; done
