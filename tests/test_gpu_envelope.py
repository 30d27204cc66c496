"""north_star's own tolerances, at BASELINE.json configs[1]'s full size, against the OTHER admissible
readings of the reference's text.

Every other GPU test demands bit equality with the default oracle, whose arithmetic choices are the
build's.  Here the oracle runs the shader's text the other way at each of those points
(oracle/rt_oracle.c, rto_set_variant): true division in the slab test (raytracer.fs.glsl:425-442), the
C library's sin/cos/acos/exp/log in place of the polynomial routines, the doHitBackside carry-over of
:522, 605-610 - and the HIP path is held to what north_star states:

  * primary-ray first-hit triangle ids equal, excluding epsilon-grazing rays (second
    candidate within 1e-5 max(1,t), |N.C| <= 1e-6 |N|^2, |N.d| < 1e-5 |N|; the mask is computed by
    the oracle in both readings);
  * accumulated radiance: per-pixel relative error <= 1e-3 at matched seeds (share of pixels stated
    and bounded: a path that meets another surface after a last-bit change diverges),
    PSNR >= 45 dB on clip(mean, 0, 1).

The measured table (all variants, also against a 4x-spp image) is profiles/r02_envelope.txt
(tools/envelope.py)."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
import envelope as E  # noqa: E402
import synth  # noqa: E402
from conftest import build_suzanne  # noqa: E402

pytestmark = pytest.mark.gpu

W, H, PASSES, BOUNCES = 1920, 1080, 64, 4
LITERAL = (1, 1, 0)           # true division + libm: the shader's text as written, as far as C can say it


def test_first_hit_and_radiance_against_the_literal_reading_full_size(oracle_lib, hip_lib):
    g = E.make(hip_lib, build_suzanne(0), W, H, sah=True, cull=True)      # the mode bench.py measures
    ids_g, ts_g = g.first_hit(1)
    img_g = E.radiance(g, PASSES, 1, BOUNCES)
    g.close()
    o = E.make(oracle_lib, build_suzanne(0), W, H, variant=LITERAL)
    ids_v, ts_v = o.first_hit(1)
    diff = ids_v != ids_g
    graz = np.zeros_like(diff)
    if diff.any():
        o0 = E.make(oracle_lib, build_suzanne(0), W, H)
        graz = diff & ((E.grazing(oracle_lib, o, only=diff) != 0) | (E.grazing(oracle_lib, o0, only=diff) != 0))
        o0.close()
    print(f"first hit vs literal reading, {W}x{H}: {int(diff.sum())} ids differ, {int(graz.sum())} of them epsilon-grazing, "
          f"{int((diff & ~graz).sum())} not")
    assert int((diff & ~graz).sum()) == 0, "a first-hit id differs on a ray that is not epsilon-grazing"
    assert diff.sum() <= 20                                               # and grazing rays are rare (measured: 0)
    same = ~diff & (ids_g >= 0)
    rel = np.abs(ts_v[same].astype(np.float64) - ts_g[same]) / ts_g[same]
    assert rel.max() <= 1e-6                                              # same triangle: distance to the last bits

    img_v = E.radiance(o, PASSES, 1, BOUNCES)
    o.close()
    bad = E.beyond_tolerance(img_g, img_v)
    p = E.psnr(img_g, img_v)
    print(f"radiance vs literal reading, {PASSES} passes: {int((img_g != img_v).any(axis=2).sum())} pixels differ in some bit, "
          f"{int(bad.sum())} beyond 1e-3 ({1 - bad.mean():.6f} within), PSNR {p:.2f} dB")
    assert bad.mean() <= 1e-3, "more than 0.1 % of the pixels beyond north_star's 1e-3 (SURVEY.md 9.7: investigate)"
    assert p >= 45.0


def test_carry_over_reading_on_the_glass_scene(oracle_lib, hip_lib):
    w, h, passes, bounces = 640, 360, 32, 8
    g = E.make(hip_lib, synth.stress_scene(), w, h)
    img_g = E.radiance(g, passes, 1, bounces)
    g.close()
    o = E.make(oracle_lib, synth.stress_scene(), w, h, variant=(1, 1, 1))
    img_v = E.radiance(o, passes, 1, bounces)
    o.close()
    bad = E.beyond_tolerance(img_g, img_v)
    p = E.psnr(img_g, img_v)
    print(f"glass scene vs carry-over + literal arithmetic: {int((img_g != img_v).any(axis=2).sum())} of {w*h} pixels differ in some bit, "
          f"{int(bad.sum())} beyond 1e-3, PSNR {p:.2f} dB")
    assert bad.mean() <= 5e-3
    assert p >= 45.0
