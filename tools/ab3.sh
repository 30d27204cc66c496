#!/bin/bash
# A/B timing of kernel variants on the three bench workloads (kernel time by HIP events, tools/prof_run.py):
# every ab/*.so (or $LIBS): Suzanne 1080p x 64 passes, dragon scene 1080p x 64 passes x 8 bounces, 991 k triangles at 4K x ${P4K:-32} passes;
# SAH structure + culling, $RUNS runs each (default 2).
for lib in ${LIBS:-ab/*.so}; do
  for i in $(seq ${RUNS:-2}); do
    echo -n "$lib  "; RT_LIB=$lib RT_SAH=1 RT_CULL=1 timeout -k 10 120 python tools/prof_run.py 64 | tail -n 1 || exit 1
  done
  for i in $(seq ${RUNS:-2}); do
    echo -n "$lib  "; RT_LIB=$lib RT_SAH=1 RT_CULL=1 RT_SCENE=dragon timeout -k 10 120 python tools/prof_run.py 64 1920 1080 8 | tail -n 1 || exit 1
  done
  if [ -z "$NO4K" ]; then
    echo -n "$lib  "; RT_LIB=$lib RT_SAH=1 RT_CULL=1 RT_SCENE=suzanne5 timeout -k 10 300 python tools/prof_run.py ${P4K:-32} 3840 2160 | tail -n 1 || exit 1
  fi
done
