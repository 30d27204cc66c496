#!/usr/bin/env python3
"""One k_render launch of bench.py's workload (optionally fewer passes) for rocprofv3 runs.
usage: python tools/prof_run.py [passes] [width height] [bounces]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import final_scene_numba as S

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
bounces = int(sys.argv[4]) if len(sys.argv) > 4 else 4
sc = S.build()
sc.animate_frame(0)
ctx = ffi.RtLib().create(0)
ctx.load_scene(sc)
ctx.load_camera(sc.cam)
ctx.resize(w, h)
import os
ctx.set_cull(os.environ.get('RT_CULL', '0') == '1')
ctx.render(1, passes, 1, 1, bounces, 0)
ctx.sync()
st = ctx.stats()
print(f"passes={passes} {w}x{h} bounces={bounces} kernel_ms={st['kernel_ms']:.3f}")
