#!/bin/bash
# The committed evidence for one bench.py configuration: kernel stats + HBM counters + wave-state counters.
#   tools/profile_bench.sh <tag> [bench.py arguments]        (through gpurun, from the repo root)
tag=$1; shift
PK_PASSES="fetch write sq1 sq2 sq3" tools/profile_kernel.sh $tag bench.py --no-cpu-baseline "$@"
