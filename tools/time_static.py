#!/usr/bin/env python3
"""What a small radiance buffer costs: bench.py's frame through rt_render (one launch, 1.6 GB of per-(pass, pixel)
radiance) and through rt_render_static (48 MiB of it: 2 passes per launch, 32 launch pairs)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import final_scene_numba as S

lib = ffi.RtLib()
sc = S.build(); sc.animate_frame(0)
ctx = lib.create(0)
ctx.set_accel(True); ctx.load_scene(sc); ctx.load_camera(sc.cam); ctx.resize(1920, 1080); ctx.set_cull(True)
for name, fn in (("rt_render", ctx.render), ("rt_render_static", ctx.render_static)):
    fn(1, 64, 1, 1, 4, 0); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.reset_accum(); fn(1, 64, 1, 1, 4, 0)
    ctx.sync()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per frame (1920x1080, 64 passes)")
