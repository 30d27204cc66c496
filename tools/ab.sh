#!/bin/bash
# A/B timing of kernel variants on the GPU box: every ab/*.so (or $LIBS) on bench.py's frame, SAH + culled,
# 3 runs each (kernel time by HIP events), then optionally the reference structure.
for lib in ${LIBS:-ab/*.so}; do
  for i in 1 2 3; do
    echo -n "$lib  "; RT_LIB=$lib RT_SAH=1 RT_CULL=1 timeout -k 10 120 python tools/prof_run.py ${PASSES:-64} | tail -n 1 || exit 1
  done
  if [ -n "$REF" ]; then echo -n "$lib ref "; RT_LIB=$lib RT_CULL=1 timeout -k 10 120 python tools/prof_run.py ${PASSES:-64} | tail -n 1 || exit 1; fi
done
