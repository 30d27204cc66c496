#!/usr/bin/env python3
"""Not collected by pytest (half a minute of 16 CPU threads; run by hand on a GPU box): BASELINE.json
configs[1] at its FULL size - Suzanne room, 1920x1080, 64 passes x 1 ray per pixel, 4 bounces - rendered by the
CPU oracle and by the HIP library in every mode; first-hit ids, distances and the float bit patterns of the
mean image must be identical, and so must the ray / triangle-test / hit counters of the exact mode.
usage: python tests/full_size_oracle.py [passes]
       RT_SCENE=dragon    : configs[4], the 192-mesh dragon scene, 1920x1080, 8 bounces
       RT_SCENE=suzanne5  : configs[3], 991 316 triangles, 3840x2160, 4 bounces (give a small pass count: the
                            oracle does ~20 Mrays/s)"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import final_scene_numba as S

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = os.environ.get("RT_SCENE", "suzanne")
W, H, BOUNCES = {"suzanne": (1920, 1080, 4), "dragon": (1920, 1080, 8), "suzanne5": (3840, 2160, 4)}[which]
_hip_for_build = ffi.RtLib()


def build_scene():
    if which == "dragon":
        from raytracer_amd.scripts.scenes import dragon_scene as D
        return D.build()
    if which == "suzanne5":
        return S.build(subdivide=5, lib=_hip_for_build)      # the native builder (rt_bvh_build): host code of the library
    sc = S.build()
    sc.animate_frame(int(os.environ.get("RT_FRAME", "0")))      # configs[2]: any frame of the keyframe animation
    return sc


def render(lib, sah, cull):
    sc = build_scene()
    c = lib.create(0)
    c.set_accel(sah)
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.resize(W, H)
    c.set_cull(cull)
    c.enable_stats(True)
    t0 = time.perf_counter()
    ids, ts = c.first_hit(1)
    c.render(1, passes, 1, 1, BOUNCES, 0)
    img = c.read_mean()
    dt = time.perf_counter() - t0
    st = c.stats()
    c.close()
    return ids, ts, img, st, dt


ref = render(ffi.RtLib(ROOT / "oracle" / "librt_oracle.so"), False, False)
print(f"oracle: {ref[3]['rays']} rays in {ref[4]:.1f} s", flush=True)
hip = ffi.RtLib()
ok = True
for name, sah, cull in (("exact", False, False), ("culled", False, True), ("SAH culled", True, True)):
    r = render(hip, sah, cull)
    d_ids = int((ref[0] != r[0]).sum()) + int((ref[1].view(np.uint32) != r[1].view(np.uint32)).sum())
    d_px = int((ref[2].view(np.uint32) != r[2].view(np.uint32)).any(axis=2).sum())
    same_counts = all(ref[3][k] == r[3][k] for k in ("rays", "shaded_hits", "samples")) and (name != "exact" or ref[3]["tri_tests"] == r[3]["tri_tests"])
    ok = ok and d_ids == 0 and d_px == 0 and same_counts
    print(f"HIP {name:10s}: {d_ids} first hits differ, {d_px} of {W*H} pixels differ, counters {'equal' if same_counts else 'DIFFER'} "
          f"(rays {r[3]['rays']}, triangle tests {r[3]['tri_tests']})", flush=True)
print(f"scene {which} frame {os.environ.get('RT_FRAME', '0')}, {W}x{H}, {passes} passes, {BOUNCES} bounces: " + ("HIP == oracle bit for bit in every mode" if ok else "MISMATCH"))
sys.exit(0 if ok else 1)
