#!/bin/bash
# quick GPU loop: parity tests, then timing of strict / cull at 64 passes
set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4
for r in ${REFILLS:-40}; do
  echo "refill_below=$r"; RT_REFILL_BELOW=$r timeout -k 10 120 python tools/prof_run.py 64
  RT_REFILL_BELOW=$r RT_CULL=1 timeout -k 10 120 python tools/prof_run.py 64
done
