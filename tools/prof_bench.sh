#!/bin/bash
# rocprofv3 evidence for bench.py's own command line (run on the GPU box, from the repo root):
#   tools/prof_bench.sh <outdir> [steps] [workload] [extra PMC sets: sq1 sq2 tcp1 ...]
# 1. kernel trace + stats of `python3 bench.py --workload W --steps S --warmup 1 --no-cpu-baseline --no-other-modes`
# 2. FETCH_SIZE and WRITE_SIZE (+ TCC hit/miss) in separate --pmc passes (3 + 2 TCC counters; never with a trace)
# 3. optional SQ / TCP passes for the same command (each its own run)
set -u
OUT=${1:-gpurun_out/prof_bench}; STEPS=${2:-3}; WL=${3:-suzanne1080}; shift 3 2>/dev/null || shift $#
SETS=$*
mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-other-modes"
echo "$BENCH" > "$OUT/command.txt"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
pmc() { name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
for s in $SETS; do case $s in
 sq1) pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY;;
 sq2) pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA;;
 sq3) pmc sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT;;
 tcp1) pmc tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum;;
esac; done
python3 tools/prof_bench_summary.py "$OUT"
