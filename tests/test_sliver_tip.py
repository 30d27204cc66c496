"""DESIGN.md 5, "sliver tips": the reference's rayTriangle (restated in oracle/np_ref.py) accepts points BEYOND the tip of a thin
triangle - by more than the SAH structure's slack alone covers - because the rounding error of an edge function is divided by
the sine of the tip's angle.  The pad that every SAH box carries since round 3 (rt_hip.hip::sliver_tip_pad, added to the slack by
k_pack_nodes) covers it.  CPU only: this pins the arithmetic fact the pad exists for and the formula's margin."""
import numpy as np

from oracle import np_ref as R

F = np.float32


def _sliver(tan_theta, rot=0.7):
    c, s = np.cos(rot), np.sin(rot)
    rot_m = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    a = np.array([0.3, 0.2, 0.0])
    tri = np.zeros(28, dtype=F)
    tri[0:3] = a
    tri[4:7] = a + rot_m @ np.array([1.0, 0.0, 0.0])
    tri[8:11] = a + rot_m @ np.array([1.0, tan_theta, 0.0])
    return tri, rot_m


def _pad_as_the_kernel_computes_it(tri):
    a, b, c = tri[0:3], tri[4:7], tri[8:11]
    n = R.cross(b - a, c - a)
    l = sorted(float(np.sum((p - q).astype(F) ** 2, dtype=F)) for p, q in ((b, a), (c, b), (a, c)))
    n2 = float(np.sum(n * n, dtype=F))
    d = np.sqrt(F(l[2]))
    return float(F(2.0 ** -19) * d * np.sqrt(F(l[2]) * F(l[1]) / F(n2)))


def test_points_beyond_a_sliver_tip_are_accepted_and_the_pad_covers_them():
    tri, rot_m = _sliver(2e-4)
    lo = np.minimum(np.minimum(tri[0:3], tri[4:7]), tri[8:11])
    hi = np.maximum(np.maximum(tri[0:3], tri[4:7]), tri[8:11])
    back, side = rot_m @ np.array([-1.0, 0.0, 0.0]), rot_m @ np.array([0.0, 1.0, 0.0])
    d = np.array([0.0, 0.0, -1.0], dtype=F)
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(12000):
        r = 10.0 ** rng.uniform(-6.0, -3.5)
        p = tri[0:3].astype(np.float64) + back * r + side * rng.uniform(-1.0, 1.0) * r * 1e-3
        o = np.array([p[0], p[1], 1.0], dtype=F)
        h = R.ray_triangle(o, d, tri, True)
        if h is not None:
            worst = max(worst, float(max(np.max(lo - h["hitPoint"]), np.max(h["hitPoint"] - hi))))
    m = float(np.max(np.abs(np.concatenate([lo, hi]))))
    widened_without_pad = 2.0 ** -16 * m + 2.0 ** -20 + 2.0 ** -17 * 1.0        # rounds 1-2: e2 / 2 + 2^-17 max|o| (now a quarter of it)
    pad = _pad_as_the_kernel_computes_it(tri)
    assert worst > widened_without_pad, (worst, widened_without_pad)            # the gap is real ...
    assert worst < 0.05 * pad, (worst, pad)                                    # ... and the pad covers it 20 times over


def test_pad_of_an_ordinary_triangle_is_negligible():
    tri = np.zeros(28, dtype=F)
    tri[0:3], tri[4:7], tri[8:11] = (0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.3, 0.8, 0.0)
    assert _pad_as_the_kernel_computes_it(tri) < 1e-5
