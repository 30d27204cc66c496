#!/bin/bash
# In-process A/B of kernel variants (ab/*.so or $LIBS): per library one process, the workload's frame rendered 1 + 3 times
# per repetition (tools/exp_env.py; kernel time by HIP events, run-to-run spread ~0.05 ms), image hash printed.
#   WL=suzanne1080|dragon|subdiv4k PASSES=64 REPS=2 tools/ab_inproc.sh
for lib in ${LIBS:-ab/*.so}; do
  RT_LIB=$lib timeout -k 10 ${TMO:-300} python tools/exp_env.py ${WL:-suzanne1080} ${PASSES:-64} RT_AB_REP $(seq ${REPS:-2}) || exit 1
done
