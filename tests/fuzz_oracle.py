#!/usr/bin/env python3
"""Not collected by pytest (run by hand on a GPU box): many more seeds of test_gpu_fuzz's random scenes,
HIP exact traversal against the CPU oracle - first-hit ids, distances, float bit patterns of the image.
usage: python tests/fuzz_oracle.py [first_seed] [n_seeds] [width height]"""
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
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (96, 54)
hip = ffi.RtLib()
oracle = ffi.RtLib(ROOT / "oracle" / "librt_oracle.so")


def render(lib, seed):
    sc = random_scene(seed)
    c = lib.create(0)
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.resize(w, h)
    ids, ts = c.first_hit(1)
    c.render(1, 2, 1, 2, 6, 0)
    img = c.read_mean()
    c.close()
    return ids, ts, img


bad = 0
for seed in range(first, first + count):
    a, b = render(oracle, seed), render(hip, seed)
    d_ids = int((a[0] != b[0]).sum()) + int((a[1].view(np.uint32) != b[1].view(np.uint32)).sum())
    d_px = int((a[2].view(np.uint32) != b[2].view(np.uint32)).any(axis=2).sum())
    if d_ids or d_px:
        bad += 1
        print(f"seed {seed}: {d_ids} first hits, {d_px} pixels differ", flush=True)
    if (seed - first) % 20 == 19:
        print(f"... {seed - first + 1} scenes", flush=True)
print(f"{count} random scenes from seed {first}, {w}x{h}, 2 passes x 2 rays, 6 bounces, HIP exact traversal vs CPU oracle: {bad} scenes with differences")
