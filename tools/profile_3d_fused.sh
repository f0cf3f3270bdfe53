#!/bin/bash
# rocprofv3 kernel stats + HBM/busy counters (separate --pmc passes) of the 3D fused kernels; run through gpurun
# from the repo root.  Output: gpurun_out/p3f_<tag>_{stats,fetch,write,busy,valu}
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
run() {  # tag, bench args...
    tag=$1; shift
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/p3f_${tag}_stats -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline "$@" > $ROOT/gpurun_out/p3f_${tag}_bench.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/p3f_${tag}_$c -- python3 $ROOT/bench.py --steps 8 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
    done
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/p3f_${tag}_busy -- python3 $ROOT/bench.py --steps 8 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $ROOT/gpurun_out/p3f_${tag}_valu -- python3 $ROOT/bench.py --steps 8 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
    cd $ROOT
    tail -1 gpurun_out/p3f_${tag}_bench.log | cut -c1-120
}
if [ "$1" = "bf16" ]; then run bf16box768 --dtype bf16; exit 0; fi
run star512 --shape star3d1r
run box768 --shape box3d1r
