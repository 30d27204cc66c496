#!/usr/bin/env python3
"""Device memory across 120 scene reloads on one context (both structures, culling on and off) and across 60
context lifetimes: hipMemGetInfo before and after.  usage: python tools/leak_check.py  (on a GPU box)"""
import sys, ctypes
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
from raytracer_amd import ffi
from test_gpu_fuzz import random_scene
hip = ctypes.CDLL("libamdhip64.so")
def free_mem():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value
lib = ffi.RtLib()
c = lib.create(0)
c.resize(320, 180)
base = None
for it in range(120):
    sc = random_scene(1 + it % 30)
    c.set_accel(it % 2 == 1)
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.set_cull(it % 3 == 0)
    c.reset_accum()
    c.render(1, 2, 1, 1, 4, 0)
    c.first_hit(1)
    c.sync()
    if it == 29: base = free_mem()
    if it % 30 == 29: print(f"iteration {it+1}: free device memory {free_mem()/2**20:.1f} MiB", flush=True)
end = free_mem()
print("same context, 90 further scene loads: free memory changed by", (end - base) / 2**20, "MiB")
c.close()
b0 = free_mem()
for it in range(60):
    c = lib.create(0); c.resize(640, 360); sc = random_scene(1 + it % 10); c.set_accel(it % 2 == 0); c.load_scene(sc); c.load_camera(sc.cam); c.render(1, 2, 1, 1, 4, 0); c.sync(); c.close()
print("60 x create/render/close: free memory changed by", (free_mem() - b0) / 2**20, "MiB")
