"""Full BASELINE.json size (1920x1080, 64 passes, 4 bounces) checks on the GPU: the whole frame against
the CPU oracle (6.4e8 queries, about 27 s of 16 CPU threads) in the exact mode and in the mode bench.py
measures, first hit against the oracle, CULL against STRICT traversal, and size-independent properties.
tests/full_size_oracle.py is the whole-frame check as a script (all modes, other configs)."""
import numpy as np
import pytest

from conftest import build_suzanne, make_ctx

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


def test_first_hit_1080p_bit_exact(oracle_lib, hip_lib):
    o = make_ctx(oracle_lib, build_suzanne(0), W, H)
    io, to = o.first_hit(1)
    for cull in (False, True):
        g = make_ctx(hip_lib, build_suzanne(0), W, H)
        g.set_cull(cull)
        ig, tg = g.first_hit(1)
        assert np.array_equal(io, ig), f"cull={cull}: {(io != ig).sum()} of {W*H} ids differ"
        assert np.array_equal(to.view(np.uint32), tg.view(np.uint32))


def test_cull_equals_strict_full_config(hip_lib):
    imgs = []
    for cull in (False, True):
        g = make_ctx(hip_lib, build_suzanne(0), W, H)
        g.set_cull(cull)
        g.render(1, 64, 1, 1, 4, 0)
        imgs.append(g.read_mean())
    diff = int((imgs[0] != imgs[1]).any(axis=2).sum())
    print(f"CULL vs STRICT at {W}x{H}x64 passes: {diff} differing pixels")
    assert diff == 0
    assert np.isfinite(imgs[0]).all() and imgs[0].max() > 0


def test_pass_split_and_sum_properties(hip_lib):
    """size-independent properties: (a) a render split into several rt_render calls equals one call
    (running mean continues exactly); (b) SUM over disjoint pass shards equals the mean within
    float summation error; (c) rendering is idempotent (no hidden state)."""
    g = make_ctx(hip_lib, build_suzanne(0), W, H)
    g.set_cull(True)
    g.render(1, 8, 1, 1, 4, 0)
    one = g.read_mean()
    g.reset_accum()
    g.render(1, 3, 1, 1, 4, 0)
    g.render(4, 5, 1, 1, 4, 0)
    assert np.array_equal(one.view(np.uint32), g.read_mean().view(np.uint32))
    shards = np.zeros_like(one, dtype=np.float64)
    for r in range(2):
        g.reset_accum()
        g.render(1 + r, 4, 2, 1, 4, 1)
        s, n = g.read_accum()
        assert n == 4
        shards += s
    assert np.abs(shards / 8 - one).max() <= 1e-5 * max(1.0, float(one.max()))
    g.reset_accum()
    g.render(1, 8, 1, 1, 4, 0)
    assert np.array_equal(one.view(np.uint32), g.read_mean().view(np.uint32))


def test_radiance_1080p_sample_rows_vs_oracle(oracle_lib, hip_lib):
    """radiance parity at full resolution on a bounded region: the oracle renders the full frame
    for 1 pass (seconds on the host cores), compared bit-for-bit."""
    o = make_ctx(oracle_lib, build_suzanne(0), W, H)
    o.render(1, 1, 1, 1, 4, 0)
    g = make_ctx(hip_lib, build_suzanne(0), W, H)
    g.set_cull(True)
    g.render(1, 1, 1, 1, 4, 0)
    a, b = o.read_mean(), g.read_mean()
    diff = int((a != b).any(axis=2).sum())
    print(f"1080p 1-pass radiance: {diff} of {W*H} pixels differ from the oracle")
    assert diff == 0


PASSES, BOUNCES = 64, 4


def _render(lib, sah, cull):
    sc = build_suzanne(0)
    c = lib.create()
    c.set_accel(sah)
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.resize(W, H)
    c.set_cull(cull)
    c.enable_stats(True)
    ids, ts = c.first_hit(1)
    c.render(1, PASSES, 1, 1, BOUNCES, 0)
    out = ids, ts, c.read_mean(), c.stats()
    c.close()
    return out


@pytest.mark.timeout(300)
def test_bench_configuration_full_size_against_the_oracle(oracle_lib, hip_lib):
    ref = _render(oracle_lib, False, False)
    assert ref[3]["rays"] == 634615464 + W * H          # the 64 passes + the first-hit probe
    for name, sah, cull in (("exact", False, False), ("SAH culled", True, True)):
        r = _render(hip_lib, sah, cull)
        assert np.array_equal(ref[0], r[0]), f"{name}: first-hit ids"
        assert np.array_equal(ref[1].view(np.uint32), r[1].view(np.uint32)), f"{name}: first-hit distances"
        assert np.array_equal(ref[2].view(np.uint32), r[2].view(np.uint32)), f"{name}: {(ref[2] != r[2]).any(axis=2).sum()} pixels differ"
        for k in ("rays", "shaded_hits", "samples"):
            assert ref[3][k] == r[3][k], (name, k)
        if name == "exact":
            assert ref[3]["tri_tests"] == r[3]["tri_tests"]
