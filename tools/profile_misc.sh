#!/bin/bash
# rocprofv3 kernel stats (+ HBM counters in separate --pmc passes) of the box2d3r fused kernel and the 1D K-step kernel;
# run through gpurun from the repo root.  Output: gpurun_out/pm_<tag>_{stats,FETCH_SIZE,WRITE_SIZE,valu}
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
run() {
    tag=$1; shift
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pm_${tag}_stats -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $ROOT/gpurun_out/pm_${tag}_bench.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pm_${tag}_$c -- python3 $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
    done
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pm_${tag}_valu -- python3 $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
    cd $ROOT
    tail -1 gpurun_out/pm_${tag}_bench.log | cut -c1-100
}
run box2d3r --shape box2d3r --steps 200
run 1d_2p20 --shape 1d1r --steps 4800 --warmup 64
