#!/bin/bash
# rocprofv3 passes over one native binary (run through gpurun from the repo root):
#   tools/profile_binary.sh <tag> <binary> [arguments]   -> gpurun_out/pb_<tag>_<pass>/ ; condense with tools/pmc_summary.py
ROOT=$(pwd)
export TMPDIR=/tmp
tag=$1; bin=$(readlink -f $2); shift 2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pb_${tag}_stats -- $bin "$@" > $ROOT/gpurun_out/pb_${tag}_stats.log 2>&1 || { echo "stats pass failed"; tail -5 $ROOT/gpurun_out/pb_${tag}_stats.log; exit 1; }
pass() {
    name=$1; shift
    case " ${PK_PASSES:-fetch write sq1 sq2 sq3} " in *" $name "*) ;; *) return 0;; esac
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/gpurun_out/pb_${tag}_$name -- $bin "${ARGS[@]}" > $ROOT/gpurun_out/pb_${tag}_$name.log 2>&1 || echo "pass $name failed"
    echo "pass $name done"
}
ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
cd $ROOT
