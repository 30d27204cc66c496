#!/usr/bin/env python3
"""The animated dragon scene (188 cubes re-posed per frame): tests per ray and kernel time of chosen frames.
usage: python tools/prof_dragon_anim.py [passes] [frames,comma,separated]   (RT_SAH=1, RT_TLAS_STATIC=1 optional)"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import dragon_scene_animated as A

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,40,70,119").split(",")]
lib = ffi.RtLib()
sc = A.build()
ctx = lib.create(0)
ctx.set_accel(os.environ.get("RT_SAH", "0") == "1")
ctx.load_scene(sc)
ctx.resize(1920, 1080)
ctx.set_cull(True)
ctx.enable_stats(True)
for f in frames:
    sc.animate_frame(f)
    ctx.update_scene(sc)
    ctx.load_camera(sc.cam)
    ctx.reset_accum()
    ctx.reset_stats()
    ctx.render(1, passes, 1, sc.cam.rays_per_pixel, sc.cam.bounces_per_ray, 0)
    ctx.sync()
    st = ctx.stats()
    r = st["rays"]
    print(f"frame {f}: rays={r} B/ray={st['box_tests']/r:.2f} T/ray={st['tri_tests']/r:.2f} kernel_ms={st['kernel_ms']:.2f}")
