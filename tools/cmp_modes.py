#!/usr/bin/env python3
"""A bench workload at its FULL size rendered in the exact mode, with both culling rules and with the band alone (SAH structure):
differing pixels between the modes, bit for bit, next to the culling audit of the exact run (rt_stats_t.cull_audit).
usage: python tools/cmp_modes.py [workload] [passes]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from raytracer_amd import ffi

name = sys.argv[1] if len(sys.argv) > 1 else "subdiv4k"
wl = bench.WORKLOADS[name]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else wl["passes"]
lib = ffi.RtLib()
scene = bench.build_scene(wl["scene"], lib)
ctx = lib.create(0)
ctx.set_accel(True)
ctx.load_scene(scene)
ctx.load_camera(scene.cam)
ctx.resize(wl["width"], wl["height"])
img = {}
for mode, label in ((0, "exact"), (1, "rules a+b"), (2, "rule a")):
    ctx.set_cull(mode)
    ctx.enable_stats(mode == 0)
    ctx.reset_stats()
    ctx.reset_accum()
    ctx.render(1, passes, 1, wl["rpp"], wl["bounces"], 0)
    img[mode] = ctx.read_mean()
    if mode == 0:
        st = ctx.stats()
        print(f"{name} {wl['width']}x{wl['height']} x {passes} passes, exact mode: {st['rays']} rays, cull_audit = {st['cull_audit']}", flush=True)
    else:
        d = (img[0].view(np.uint32) != img[mode].view(np.uint32)).any(axis=2)
        worst = float(np.abs(img[0] - img[mode]).max())
        print(f"  {label:10s}: {int(d.sum())} of {d.size} pixels differ from the exact mode (largest channel difference {worst:.3g})", flush=True)
ctx.close()
