#!/usr/bin/env python3
"""CULL vs STRICT on the bench frame: differing pixels, tests per ray, kernel time.
usage: [RT_LIB=...] python tools/cmp_cull.py [passes] [width height] [bounces]"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import final_scene_numba as S

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
bounces = int(sys.argv[4]) if len(sys.argv) > 4 else 4
frames = [int(x) for x in os.environ.get("RT_FRAMES", os.environ.get("RT_FRAME", "0")).split(",")]
lib = ffi.RtLib(os.environ.get("RT_LIB") or None)
total_diff = total_rays = 0
for frame in frames:
  sc = S.build()
  sc.animate_frame(frame)
  ctx = lib.create(0)
  ctx.load_scene(sc)
  ctx.load_camera(sc.cam)
  ctx.resize(w, h)
  out = {}
  for cull in (False, True):
      ctx.set_cull(cull)
      ctx.enable_stats(True)
      ctx.reset_stats()
      ctx.reset_accum()
      ctx.render(1, passes, 1, 1, bounces, 0)
      img = ctx.read_mean()
      st = ctx.stats()
      ctx.enable_stats(False)
      ctx.reset_stats()
      ctx.reset_accum()
      ctx.render(1, passes, 1, 1, bounces, 0)
      ctx.sync()
      ms = ctx.stats()["kernel_ms"]
      out[cull] = img
      print(f"{'CULL  ' if cull else 'STRICT'} B/ray={st['box_tests']/st['rays']:.2f} T/ray={st['tri_tests']/st['rays']:.2f} kernel_ms={ms:.2f}")
  d = (out[False].view(np.uint32) != out[True].view(np.uint32)).any(axis=2)
  print(f"frame {frame}: differing pixels: {int(d.sum())} of {w*h} ({passes} passes, {bounces} bounces)")
  total_diff += int(d.sum())
  total_rays += st["rays"]
  ctx.close()
print(f"TOTAL over {len(frames)} frames: {total_diff} differing pixels, {total_rays:.3g} rays per mode")
