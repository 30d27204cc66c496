#!/bin/bash
# quick GPU loop: parity tests, then timing of cull / strict at 64 passes for the libs in $LIBS
set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4
for lib in ${LIBS:-raytracer_amd/librt_hip.so}; do
  echo "== $lib"
  RT_LIB=$lib RT_CULL=1 timeout -k 10 120 python tools/prof_run.py 64
  RT_LIB=$lib timeout -k 10 120 python tools/prof_run.py 64
done
