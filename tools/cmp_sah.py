#!/usr/bin/env python3
"""The opt-in SAH structure against the reference structure on the bench frame: differing first
hits / pixels, tests per ray, kernel time.
usage: python tools/cmp_sah.py [passes] [width height] [bounces]
       RT_FRAMES=0,40,...  animation frames; RT_SCENE=suzanne (default) | suzanneN (N x subdivided) | dragon | dragon_animated"""
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
frames = [int(x) for x in os.environ.get("RT_FRAMES", "0").split(",")]
lib = ffi.RtLib(os.environ.get("RT_LIB") or None)
which = os.environ.get("RT_SCENE", "suzanne")


def build_scene():
    if which == "dragon":
        from raytracer_amd.scripts.scenes import dragon_scene as D
        return D.build()
    if which == "dragon_animated":
        from raytracer_amd.scripts.scenes import dragon_scene_animated as D
        return D.build()
    if which != "suzanne":
        return S.build(subdivide=int(which[7:]), lib=lib)
    return S.build()


tot_px = tot_ids = 0
for frame in frames:
    res = {}
    for name, sah, cull in (("reference strict", False, False), ("reference cull", False, True), ("SAH cull", True, True)):
        sc = build_scene()
        ctx = lib.create(0)
        ctx.set_accel(sah)
        ctx.load_scene(sc)                  # uploaded as built, then the frame's poses (the reference's order)
        sc.animate_frame(frame)
        ctx.update_scene(sc)
        ctx.load_camera(sc.cam)
        ctx.resize(w, h)
        ctx.set_cull(cull)
        ids, ts = ctx.first_hit(1)
        ctx.enable_stats(True)
        ctx.reset_stats()
        ctx.render(1, passes, 1, 1, bounces, 0)
        img = ctx.read_mean()
        st = ctx.stats()
        ctx.enable_stats(False)
        ctx.reset_stats()
        ctx.reset_accum()
        ctx.render(1, passes, 1, 1, bounces, 0)
        ctx.sync()
        ms = ctx.stats()["kernel_ms"]
        res[name] = (ids, ts, img)
        print(f"frame {frame} {name:17s} B/ray={st['box_tests']/st['rays']:.2f} T/ray={st['tri_tests']/st['rays']:.2f} rays={st['rays']} kernel_ms={ms:.2f}")
        ctx.close()
    a, b = res["reference strict"], res["SAH cull"]
    d_ids = int((a[0] != b[0]).sum())
    d_px = int((a[2].view(np.uint32) != b[2].view(np.uint32)).any(axis=2).sum())
    err = np.abs(a[2] - b[2]).max()
    print(f"frame {frame}: SAH vs reference: {d_ids} first-hit ids differ, {d_px} of {w*h} pixels differ (max abs {err:.3g})")
    tot_px += d_px
    tot_ids += d_ids
print(f"scene {which}, {w}x{h}, {passes} passes, {bounces} bounces.  TOTAL over {len(frames)} frames: {tot_ids} first-hit ids, {tot_px} pixels differ")
