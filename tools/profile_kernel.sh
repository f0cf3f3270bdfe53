#!/bin/bash
# rocprofv3 passes over one command (run through gpurun from the repo root):
#   tools/profile_kernel.sh <tag> <script.py> [arguments]        e.g.  ... r02_x bench.py --steps 20 --no-cpu-baseline
# Kernel trace + stats first, then hardware counters in passes of their own (never combined with tracing other than
# --kernel-trace).  PK_PASSES selects the counter passes (default: all).  Output: gpurun_out/pk_<tag>_<pass>/;
# condense with tools/pmc_summary.py.
ROOT=$(pwd)
export TMPDIR=/tmp
tag=$1; script=$2; shift 2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pk_${tag}_stats -- python3 $ROOT/$script "$@" > $ROOT/gpurun_out/pk_${tag}_stats.log 2>&1 || { echo "stats pass failed"; tail -5 $ROOT/gpurun_out/pk_${tag}_stats.log; exit 1; }
grep '^{' $ROOT/gpurun_out/pk_${tag}_stats.log | tail -3 | cut -c1-300
pass() {
    name=$1; shift
    case " ${PK_PASSES:-fetch write sq1 sq2 sq3 tcc vm tcp wr rd ea mfma} " in *" $name "*) ;; *) return 0;; esac
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/gpurun_out/pk_${tag}_$name -- python3 $ROOT/$script "${ARGS[@]}" > $ROOT/gpurun_out/pk_${tag}_$name.log 2>&1 || echo "pass $name failed"
    echo "pass $name done"
}
ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_BUSY_CYCLES
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass vm SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
pass ea TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
cd $ROOT
