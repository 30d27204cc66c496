#!/bin/bash
# PMC passes for k_trace (each its own run; <= 8 SQ / 4 TCC / 4 TCP counters per pass).
# usage: tools/prof_pmc.sh <outdir> [passes] [sets...]   sets: sq1 sq2 sq3 tcc1 tcc2 tcp1
set -u
OUT=${1:-gpurun_out/pmc}; PASSES=${2:-8}; shift 2
SETS=${*:-sq1 sq2 sq3 tcc1 tcc2 tcp1}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 tools/prof_run.py "$PASSES" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
for s in $SETS; do case $s in
 sq1) run sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1;;
 sq2) run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA || exit 1;;
 sq3) run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT || exit 1;;
 tcc1) run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE || exit 1;;
 tcc2) run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1;;
 tcp1) run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum || exit 1;;
esac; done
