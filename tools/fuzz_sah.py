#!/usr/bin/env python3
"""Random scenes (tests/test_gpu_fuzz.py::random_scene: soups with slivers, boxes, spheres, glass, veils,
3 - 60 posed meshes): SAH structure + culling against the reference structure's exact traversal, first
hits and float bit patterns of the image.  usage: python tools/fuzz_sah.py [first_seed] [n_seeds] [width height]
RT_FUZZ_NO_SPHERES=1 drops the analytic spheres, so that every scene of up to 32 meshes whose root table fits takes the
instantiation with the root table's octant variants (ROCT); the last line says how many did."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from raytracer_amd import ffi
from test_gpu_fuzz import random_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (192, 108)
lib = ffi.RtLib()
no_spheres = os.environ.get("RT_FUZZ_NO_SPHERES", "0") != "0"
n_roct = 0


def render(seed, sah, cull):
    global n_roct
    sc = random_scene(seed)
    if no_spheres:
        sc.spheres.clear()
    c = lib.create(0)
    c.set_accel(sah)
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.resize(w, h)
    c.set_cull(cull)
    ids, ts = c.first_hit(1)
    c.render(1, 4, 1, 2, 6, 0)
    img = c.read_mean()
    n_roct += c.info().get("root_table_variants") == 8
    c.close()
    return ids, ts, img


bad = 0
for seed in range(first, first + count):
    a = render(seed, False, False)
    for name, sah, cull in (("reference culled", False, True), ("SAH culled", True, True)):
        b = render(seed, sah, cull)
        d_ids = int((a[0] != b[0]).sum()) + int((a[1].view(np.uint32) != b[1].view(np.uint32)).sum())
        d_px = int((a[2].view(np.uint32) != b[2].view(np.uint32)).any(axis=2).sum())
        if d_ids or d_px:
            bad += 1
            print(f"seed {seed} {name}: {d_ids} first hits, {d_px} pixels differ", flush=True)
    if (seed - first) % 20 == 19:
        print(f"... {seed - first + 1} scenes", flush=True)
print(f"{count} random scenes from seed {first}{' without spheres' if no_spheres else ''}, {w}x{h}, 4 passes x 2 rays, 6 bounces: {bad} comparisons with differences "
      f"({n_roct} SAH renders read the root table's octant variants)")
