#!/bin/bash
# What one more instruction costs where (profiles/r03_k_trace_isa.txt): builds the working tree with N dummy
# instructions at the END of a block (rt_dev_trace.h::cal_valu / cal_salu) and times bench.py's frame with each library.
#   CPU box:  tools/calibrate.sh build        GPU box:  tools/calibrate.sh run   (WL=suzanne1080|dragon|subdiv4k PASSES=64)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  tools/build_variant.sh cal_base -
  tools/build_variant.sh cal_nodeV8 - -DRT_CAL_NODE_VALU=8
  tools/build_variant.sh cal_nodeS16 - -DRT_CAL_NODE_SALU=16
  tools/build_variant.sh cal_triV8 - -DRT_CAL_TRI_VALU=8
  tools/build_variant.sh cal_pathV64 - -DRT_CAL_PATH_VALU=64
else
  LIBS="ab/cal_base.so ab/cal_nodeV8.so ab/cal_nodeS16.so ab/cal_triV8.so ab/cal_pathV64.so ab/cal_base.so" REPS=1 bash tools/ab_inproc.sh
fi
