#!/usr/bin/env python3
"""Distance between the product and the OTHER admissible readings of the reference's shader text
(VERDICT r1, item 1), measured with north_star's own tolerances.  GPU box only.

The product is bit-identical to the default oracle - which fixes, as DESIGN.md 2 says, the points
GLSL leaves to the driver one way.  oracle/rt_oracle.c can also be run the other way at each of
them (rto_set_variant): true division in the slab test (raytracer.fs.glsl:425-442 as written), the
C library's sinf/cosf/acosf/expf/logf where the GL driver's functions would run, and the
`doHitBackside` carry-over of :522, 605-610.  This script renders BASELINE.json configs[1]
(Suzanne room, 1920x1080, 64 passes x 1 ray, 4 bounces) with the HIP path and with each variant and
reports

  * first hit: differing triangle ids, how many of them are epsilon-grazing
    (second candidate within 1e-5 max(1, t), |N.C| <= 1e-6 |N|^2, |N.d| < 1e-5 |N|; mask taken in
    BOTH readings), how many are not;
  * radiance at matched seeds: pixels differing in any bit, pixels beyond
    |a - b| <= 1e-3 max(|b|, 1e-3 mean(b)), PSNR on clip(mean, 0, 1) against the variant at equal
    spp and against a 4x-spp image of the variant (paths that diverge do so chaotically: the
    second figure says whether the two 64-pass images are equally good estimates of one image);
  * the carry-over variant on the glass scene of tests/synth.py (rays inside solids).

usage: python tools/envelope.py [--width 1920 --height 1080 --passes 64 --hi-factor 4] > profiles/r02_envelope.txt
"""
from __future__ import annotations

import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from raytracer_amd import ffi  # noqa: E402


def psnr(a, b):
    a, b = np.clip(a, 0, 1).astype(np.float64), np.clip(b, 0, 1).astype(np.float64)
    mse = float(np.mean((a - b) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


def beyond_tolerance(a, b):
    """pixels with some channel beyond |a-b| <= 1e-3 max(|b|, 1e-3 mean(b))  """
    tol = 1e-3 * np.maximum(np.abs(b), 1e-3 * float(b.mean()))
    return (np.abs(a - b) > tol).any(axis=2)


def grazing(lib, ctx, only=None, pass_index=1):
    lib.dll.rto_first_hit_grazing.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    m = np.zeros((ctx.height, ctx.width), dtype=np.uint8)
    o = None if only is None else np.ascontiguousarray(only, dtype=np.uint8)
    rc = lib.dll.rto_first_hit_grazing(ctx._h, pass_index, m.ctypes.data, None if o is None else o.ctypes.data)
    assert rc == 0
    return m


def make(lib, scene, w, h, variant=None, sah=False, cull=False):
    ctx = lib.create(0)
    if lib.backend == "hip":
        ctx.set_accel(sah)
    ctx.load_scene(scene)
    ctx.load_camera(scene.cam)
    ctx.resize(w, h)
    if lib.backend == "hip":
        ctx.set_cull(cull)
    if variant is not None:
        lib.dll.rto_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        assert lib.dll.rto_set_variant(ctx._h, *variant) == 0
    return ctx


def radiance(ctx, passes, rpp, bounces, first=1):
    ctx.reset_accum()
    ctx.render(first, passes, 1, rpp, bounces, ffi.RT_ACCUM_MEAN)
    return ctx.read_mean()


VARIANTS = [("true division", (1, 0, 0)), ("libm functions", (0, 1, 0)), ("both = the literal text", (1, 1, 0))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--passes", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--hi-factor", type=int, default=4)
    ap.add_argument("--glass-size", type=int, nargs=2, default=(960, 540))
    args = ap.parse_args()
    from conftest import build_suzanne
    import synth

    W, H, P, B = args.width, args.height, args.passes, args.bounces
    hip = ffi.RtLib()
    orc = ffi.RtLib(ROOT / "oracle" / "librt_oracle.so")
    print(f"# envelope of the arithmetic choices, Suzanne room {W}x{H}, {P} passes x 1 ray/pixel, {B} bounces, camera keyframe 0")
    print(f"# product = librt_hip.so ({hip.backend}); variants = oracle/rt_oracle.c with rto_set_variant(true_div, libm, carry)")

    t0 = time.time()
    g = make(hip, build_suzanne(0), W, H)
    ids_g, ts_g = g.first_hit(1)
    img_g = radiance(g, P, 1, B)
    g.close()
    gs = make(hip, build_suzanne(0), W, H, sah=True, cull=True)
    img_gs = radiance(gs, P, 1, B)
    gs.close()
    print(f"# HIP exact mode vs HIP SAH+culled (the bench mode): {int((img_g != img_gs).any(axis=2).sum())} pixels differ")

    o0 = make(orc, build_suzanne(0), W, H)
    ids_0, ts_0 = o0.first_hit(1)
    print(f"# default oracle vs HIP, first hit: {int((ids_0 != ids_g).sum())} ids, {int((ts_0.view(np.uint32) != ts_g.view(np.uint32)).sum())} distances differ "
          f"(bit-exact parity is the subject of tests/test_gpu_fullsize.py)")
    m0_full = grazing(orc, o0)
    n_px = W * H
    print(f"# epsilon-grazing primary rays in the default reading: {int((m0_full != 0).sum())} of {n_px} "
          f"(near tie {int((m0_full & 1 != 0).sum())}, edge {int((m0_full & 2 != 0).sum())}, parallel {int((m0_full & 4 != 0).sum())}, near miss {int((m0_full & 8 != 0).sum())})")

    print("\n## first hit, pass 1 (ids of HIP vs variant)")
    print("variant | ids differing | of which grazing (either reading) | NOT grazing | distances differing in some bit | max |dt|/t")
    for name, v in VARIANTS:
        o = make(orc, build_suzanne(0), W, H, variant=v)
        ids_v, ts_v = o.first_hit(1)
        diff = ids_v != ids_g
        mv = grazing(orc, o, only=diff)
        graz = diff & ((mv != 0) | (m0_full != 0))
        same = ~diff & (ids_g >= 0)
        dt = np.abs(ts_v[same].astype(np.float64) - ts_g[same]) / np.maximum(ts_g[same], 1e-30)
        print(f"{name} | {int(diff.sum())} | {int(graz.sum())} | {int((diff & ~graz).sum())} | "
              f"{int((ts_v.view(np.uint32) != ts_g.view(np.uint32)).sum())} | {dt.max() if dt.size else 0:.3g}")
        o.close()

    print(f"\n## radiance, passes 1..{P} (running mean), HIP vs variant at matched seeds")
    print("variant | pixels differing in some bit | pixels beyond 1e-3 | share within 1e-3 | PSNR(HIP, variant) dB | "
          f"PSNR(HIP, variant at {args.hi_factor}x spp) dB | PSNR(variant, variant at {args.hi_factor}x spp) dB")
    for name, v in VARIANTS:
        o = make(orc, build_suzanne(0), W, H, variant=v)
        img_v = radiance(o, P, 1, B)
        hi = None
        if v == (1, 1, 0) and args.hi_factor > 1:
            hi = radiance(o, P * args.hi_factor, 1, B)
        bad = beyond_tolerance(img_g, img_v)
        line = (f"{name} | {int((img_g != img_v).any(axis=2).sum())} | {int(bad.sum())} | {1.0 - bad.mean():.6f} | {psnr(img_g, img_v):.2f}")
        if hi is not None:
            line += f" | {psnr(img_g, hi):.2f} | {psnr(img_v, hi):.2f}"
        else:
            line += " | - | -"
        print(line, flush=True)
        o.close()
    o0.close()

    gw, gh = args.glass_size
    print(f"\n## doHitBackside carry-over (:522, 605-610), glass scene of tests/synth.py (40 cubes, glass ball, spheres), {gw}x{gh}, 16 passes, 8 bounces")
    sg = make(hip, synth.stress_scene(), gw, gh)
    img_sg = radiance(sg, 16, 1, 8)
    sg.close()
    for name, v in (("default (no carry-over)", (0, 0, 0)), ("carry-over", (0, 0, 1)), ("carry-over + literal arithmetic", (1, 1, 1))):
        o = make(orc, synth.stress_scene(), gw, gh, variant=v)
        img_v = radiance(o, 16, 1, 8)
        bad = beyond_tolerance(img_sg, img_v)
        print(f"{name} | pixels differing in some bit {int((img_sg != img_v).any(axis=2).sum())} of {gw*gh} | beyond 1e-3: {int(bad.sum())} "
              f"| share within {1.0 - bad.mean():.6f} | PSNR {psnr(img_sg, img_v):.2f} dB", flush=True)
        o.close()
    print(f"\n# wall time {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
