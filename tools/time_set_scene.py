#!/usr/bin/env python3
"""Host-side set-up times of the 991 316-triangle scene: scene build (OBJ, subdivision, native reference BVH)
and rt_set_scene with and without the SAH build.  usage: python tools/time_set_scene.py"""
import sys, time
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import final_scene_numba as S
lib = ffi.RtLib()
t0 = time.perf_counter(); sc = S.build(subdivide=5, lib=lib); t1 = time.perf_counter()
print(f"scene build (OBJ + subdivision + native reference BVH): {t1-t0:.2f} s, {sc.count_triangles()} triangles")
for sah in (False, True):
    ctx = lib.create(0); ctx.set_accel(sah)
    t0 = time.perf_counter(); ctx.load_scene(sc); ctx.sync(); t1 = time.perf_counter()
    print(f"load_scene sah={sah}: {t1-t0:.2f} s")
    ctx.close()
