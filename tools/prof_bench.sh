#!/bin/bash
# rocprofv3 evidence for bench.py's own command line (run on the GPU box, from the repo root):
#   tools/prof_bench.sh <outdir> [steps]
# 1. kernel trace + stats of `python3 bench.py --steps S --warmup 1 --no-cpu-baseline`
# 2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (3 + 2 TCC counters; never with a trace)
set -u
OUT=${1:-gpurun_out/prof_bench}; STEPS=${2:-3}
mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $BENCH > "$OUT/fetch.log" 2>&1 || { echo "fetch failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/write" -o write -- $BENCH > "$OUT/write.log" 2>&1 || { echo "write failed"; tail -5 "$OUT/write.log"; exit 1; }
python3 tools/prof_bench_summary.py "$OUT"
