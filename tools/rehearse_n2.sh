#!/bin/bash
# Walk the N > 1 code paths with two ranks sharing the one GPU of the box (gloo between the ranks; tiny frames:
# two chip-filling persistent kernels time-slice badly on one card).  NOT a measurement.
set -e; echo "rehearsal start"
export RT_BENCH_REHEARSAL=1 RT_BENCH_BACKEND=gloo RT_BENCH_WATCHDOG=${RT_BENCH_WATCHDOG:-100}
for sc in weak strong; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 2 --warmup 1 --scaling $sc
done
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
  -m raytracer_amd.distributed --frames 6 --height 90 --passes 4 --gather
