#!/bin/bash
# The committed evidence for every BASELINE config: per config the bench.py line, rocprofv3 kernel stats and the counter
# passes (fetch, write, sq1, sq2, sq3), condensed into profiles/<round>_<name>_{kernel_stats.csv,pmc.json} and the
# profiles/pmc_traffic.json entry bench.py cites.   tools/profile_configs.sh r04    (through gpurun, from the repo root)
round=${1:-r04}
# PC_ONLY="star3d1r box3d1r": only these configs
run() {
    name=$1; kernel=$2; shift 2
    if [ -n "$PC_ONLY" ]; then case " $PC_ONLY " in *" $name "*) ;; *) return 0;; esac; fi
    tag=${round}_${name}
    tools/profile_bench.sh $tag "$@" > gpurun_out/pc_${tag}.log 2>&1
    key=$(grep '^{' gpurun_out/pk_${tag}_stats.log | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["roofline"]["traffic_key"])')
    python3 tools/pmc_summary.py --tag $tag --stats gpurun_out/pk_${tag}_stats --pmc gpurun_out/pk_${tag}_fetch gpurun_out/pk_${tag}_write \
        gpurun_out/pk_${tag}_sq1 gpurun_out/pk_${tag}_sq2 gpurun_out/pk_${tag}_sq3 --kernel "$kernel" --traffic-key "$key" >> gpurun_out/pc_${tag}.log 2>&1
    echo "$tag: $key"; tail -1 gpurun_out/pc_${tag}.log | cut -c1-300
}
run star2d1r "stencil2d_wg_kernel<7, 3" --steps 20 --warmup 5
run box2d3r "stencil2d_wg_kernel" --shape box2d3r --steps 200
run star2d3r "stencil2d_wg_kernel" --shape star2d3r --steps 100
run star3d1r "stencil3d_lanes_kernel<0, 4" --shape star3d1r --steps 50
run box3d1r "stencil3d_lanes_kernel<2, 4" --shape box3d1r --steps 50
run box3d1r_bf16 "stencil3d_bf16_lanes_kernel<4" --shape box3d1r --dtype bf16 --steps 50
run 1d1r "stencil1d_fusedk_kernel" --shape 1d1r --steps 100
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
mkdir -p gpurun_out/profiles_${round}; cp profiles/${round}_*_pmc.json profiles/${round}_*_kernel_stats.csv gpurun_out/profiles_${round}/ 2>/dev/null
