#!/usr/bin/env python3
"""One k_trace launch of bench.py's workload (optionally fewer passes) for rocprofv3 runs.
usage: python tools/prof_run.py [passes] [width height] [bounces]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import final_scene_numba as S

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
bounces = int(sys.argv[4]) if len(sys.argv) > 4 else 4
import os
lib = ffi.RtLib(os.environ.get('RT_LIB') or None)
which = os.environ.get('RT_SCENE', 'suzanne')          # suzanne | suzanne5 (991k triangles) | dragon (192 meshes)
if which == 'dragon':
    from raytracer_amd.scripts.scenes import dragon_scene as D
    sc = D.build()
elif which.startswith('suzanne') and which != 'suzanne':
    sc = S.build(subdivide=int(which[7:]), lib=lib)
else:
    sc = S.build()
sc.animate_frame(0)
ctx = lib.create(0)
ctx.set_accel(os.environ.get('RT_SAH', '0') == '1')
ctx.load_scene(sc)
ctx.load_camera(sc.cam)
ctx.resize(w, h)
import os
ctx.set_cull(os.environ.get('RT_CULL', '0') == '1')
if os.environ.get('RT_STATS') == '1':
    ctx.enable_stats(True)
ctx.render(1, passes, 1, 1, bounces, 0)
ctx.sync()
st = ctx.stats()
if os.environ.get("RT_STATS") == "1":
    r = st["rays"]; sd = st["sched"]
    print(f"rays={r} B/ray={st['box_tests']/r:.2f} T/ray={st['tri_tests']/r:.2f} H/ray={st['shaded_hits']/r:.3f}")
    print(f"node steps/ray={sd[1]/r:.2f} avg lanes={sd[1]/max(sd[0],1):.1f}; tri steps/ray={sd[3]/r:.2f} avg lanes={sd[3]/max(sd[2],1):.1f}; path steps/ray={sd[5]/r:.2f} avg lanes={sd[5]/max(sd[4],1):.1f}")
print(f"scene={which} tris={sc.count_triangles()} passes={passes} {w}x{h} bounces={bounces} kernel_ms={st['kernel_ms']:.3f}")
