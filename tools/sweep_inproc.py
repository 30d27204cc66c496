#!/usr/bin/env python3
"""Scheduler-threshold sweep in one process per workload (kernel time by HIP events, 3 timed launches per point):
RT_REFILL_BELOW x RT_BURST8 are read at rt_create, so every point is a fresh context on the same loaded scene.
usage: python tools/sweep_inproc.py [workload] [passes] [refill values, comma separated] [burst8 values]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from raytracer_amd import ffi

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "suzanne1080"]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else wl["passes"]
refills = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16,20,24").split(",")]
bursts = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "2,3,4").split(",")]
lib = ffi.RtLib(os.environ.get("RT_LIB") or None)
scene = bench.build_scene(wl["scene"], lib)
for r in refills:
    for b in bursts:
        os.environ["RT_REFILL_BELOW"], os.environ["RT_BURST8"] = str(r), str(b)
        ctx = lib.create(0)
        ctx.set_accel(True)
        ctx.load_scene(scene)
        ctx.load_camera(scene.cam)
        ctx.resize(wl["width"], wl["height"])
        ctx.set_cull(1)
        ctx.render(1, passes, 1, wl["rpp"], wl["bounces"], 0)
        ms = []
        for _ in range(3):
            ctx.reset_stats()
            ctx.reset_accum()
            ctx.render(1, passes, 1, wl["rpp"], wl["bounces"], 0)
            ctx.sync()
            ms.append(ctx.stats()["kernel_ms"])
        print(f"{sys.argv[1] if len(sys.argv) > 1 else 'suzanne1080'} refill_below={r:2d} burst8={b}: " + " ".join(f"{m:.2f}" for m in ms), flush=True)
        ctx.close()
