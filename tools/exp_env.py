#!/usr/bin/env python3
"""A/B of a structure-building knob of rt_set_scene that is read from the environment (RT_SAH_GROUP_BELOW = mesh grouping of
the SAH structure, RT_TREELET = record order, RT_SAH_LEAF ...) on a bench workload: kernel time, tests per ray, and the image
against the first value's, bit for bit.
usage: python tools/exp_env.py [workload] [passes] [VAR] [value ...]"""
import hashlib
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from raytracer_amd import ffi

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "suzanne1080"]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else wl["passes"]
var = sys.argv[3] if len(sys.argv) > 3 else "RT_SAH_GROUP_BELOW"
belows = [int(x) for x in sys.argv[4:]] or [0, 64, 1 << 30]
lib = ffi.RtLib(os.environ.get("RT_LIB") or None)
scene = bench.build_scene(wl["scene"], lib)
ref = None
for below in belows:
    os.environ[var] = str(below)
    ctx = lib.create(0)
    ctx.set_accel(True)
    ctx.load_scene(scene)
    ctx.load_camera(scene.cam)
    ctx.resize(wl["width"], wl["height"])
    ctx.set_cull(int(os.environ.get("RT_CULL", "1")))
    ctx.enable_stats(True)
    ctx.render(1, passes, 1, wl["rpp"], wl["bounces"], 0)
    st = ctx.stats()
    ctx.enable_stats(False)
    ms = []
    for _ in range(3):
        ctx.reset_stats()
        ctx.reset_accum()
        ctx.render(1, passes, 1, wl["rpp"], wl["bounces"], 0)
        ctx.sync()
        ms.append(ctx.stats()["kernel_ms"])
    img = ctx.read_mean()
    h = hashlib.sha1(img.tobytes()).hexdigest()[:12]
    ref = ref or h
    info = ctx.info()
    r = st["rays"]
    sd = st["sched"]
    print(f"{os.path.basename(os.environ.get('RT_LIB') or 'librt_hip.so')} {var}={below:<10d} roots={info['n_roots']} records={info['n_records']} stack={info['stack_entries_lds']}/{info['stack_entries']} "
          f"B/ray={st['box_tests']/r:.2f} T/ray={st['tri_tests']/r:.2f} node lane-steps/ray={sd[1]/r:.2f} tri={sd[3]/r:.2f} "
          f"ms={' '.join(f'{m:.2f}' for m in ms)} image={'same' if h == ref else 'DIFFERS'} {h}", flush=True)
    ctx.close()
