"""The opt-in SAH structure (rt_set_accel(RT_ACCEL_SAH)): same triangles, same tie ranks, other
boxes.  Its slab tests are widened so that no triangle the reference could hit is missed, and a
hit counts only if the ray passes the box of the triangle's leaf in the REFERENCE's tree, so the
set of triangles that can win is the reference's: the tests demand identical first hits and
bit-identical radiance against the reference structure's exact traversal."""
import os

import numpy as np
import pytest

from conftest import build_suzanne
from test_gpu_fuzz import random_scene
from synth import stress_scene


def _ctx(lib, sc, w, h, sah, cull):
    c = lib.create()
    c.set_accel(sah)
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.resize(w, h)
    c.set_cull(cull)
    return c


def _render(lib, sc, w, h, sah, cull, passes, rpp, bounces):
    c = _ctx(lib, sc, w, h, sah, cull)
    c.enable_stats(True)
    ids, ts = c.first_hit(1)
    c.render(1, passes, 1, rpp, bounces, 0)
    return ids, ts, c.read_mean(), c.stats()


@pytest.mark.gpu
def test_sah_on_random_scenes(hip_lib):
    tot = bad = 0
    for seed in range(1, 25):
        w, h = 96, 54
        a = _render(hip_lib, random_scene(seed), w, h, False, False, 2, 2, 6)
        b = _render(hip_lib, random_scene(seed), w, h, True, True, 2, 2, 6)
        d_ids = int((a[0] != b[0]).sum())
        d_px = int((a[2].view(np.uint32) != b[2].view(np.uint32)).any(axis=2).sum())
        tot += w * h
        bad += d_px
        assert d_ids == 0 and d_px == 0, f"seed {seed}: {d_ids} first hits, {d_px} pixels differ"
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert b[3]["tri_tests"] < a[3]["tri_tests"]
    print(f"SAH vs reference structure on 24 random scenes: {bad} of {tot} pixels differ")


@pytest.mark.gpu
def test_sah_stress_scene_and_pose_updates(hip_lib):
    a = _render(hip_lib, stress_scene(), 160, 90, False, False, 3, 2, 8)
    b = _render(hip_lib, stress_scene(), 160, 90, True, True, 3, 2, 8)
    d = int((a[2].view(np.uint32) != b[2].view(np.uint32)).any(axis=2).sum())
    print(f"stress scene: {d} of {160*90} pixels differ")
    assert d == 0 and a[3]["rays"] > 0 and np.array_equal(a[0], b[0])
    # animation: pose and material updates reach the SAH structure too (device refit of ITS boxes)
    sc0, sc1 = build_suzanne(0), build_suzanne(0)
    ref = _ctx(hip_lib, sc0, 160, 90, False, False)
    sah = _ctx(hip_lib, sc1, 160, 90, True, True)
    for frame in (0, 37, 90):
        for sc, ctx in ((sc0, ref), (sc1, sah)):
            sc.animate_frame(frame)
            ctx.update_scene(sc)
            ctx.load_camera(sc.cam)
            ctx.reset_accum()
            ctx.render(1, 2, 1, 1, 4, 0)
        assert np.array_equal(ref.read_mean().view(np.uint32), sah.read_mean().view(np.uint32)), f"frame {frame}"
    with pytest.raises(Exception):
        from raytracer_amd.scripts.scenes.chunker import NODE_DTYPE
        sah.read_nodes48(NODE_DTYPE)            # the reference's nodes are not on the device in SAH mode
    assert np.array_equal(ref.read_tris28().view(np.uint32), sah.read_tris28().view(np.uint32))


@pytest.mark.gpu
def test_sah_bench_frame_is_pixel_identical(hip_lib):
    """What bench.py relies on: on configs[1] the SAH + culled traversal renders the very image of
    the reference structure's exact traversal, first hits included."""
    w, h = 1920, 1080
    a = _render(hip_lib, build_suzanne(0), w, h, False, False, 64, 1, 4)
    b = _render(hip_lib, build_suzanne(0), w, h, True, True, 64, 1, 4)
    assert a[3]["rays"] == b[3]["rays"] == 634615464 + w * h      # 64 passes + the first-hit probe
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    print(f"triangle tests per ray {a[3]['tri_tests']/a[3]['rays']:.1f} -> {b[3]['tri_tests']/b[3]['rays']:.1f}, "
          f"box tests per ray {a[3]['box_tests']/a[3]['rays']:.1f} -> {b[3]['box_tests']/b[3]['rays']:.1f}")


@pytest.mark.gpu
def test_sah_repeat_path_gives_the_same_image(hip_lib, monkeypatch):
    """The repeat path (a winner that fails the reference's leaf box -> the query runs again with
    the check on every hit) triggers in ~1e-9 of queries; RT_SAH_RECHECK_ALL=1 (read at rt_create)
    sends every query that hit a triangle through it.  Images, first hits and ray counts stay
    the same, on the Suzanne room and on scenes with spheres, glass and leaf roots."""
    cases = [(lambda: build_suzanne(0), 320, 180, 4, 1, 4),
             (stress_scene, 160, 90, 3, 2, 8),
             (lambda: random_scene(7), 96, 54, 2, 2, 6)]
    for make, w, h, passes, rpp, bounces in cases:   # scenes are built one at a time: mesh ids are process-wide
        a = _render(hip_lib, make(), w, h, False, False, passes, rpp, bounces)
        monkeypatch.setenv("RT_SAH_RECHECK_ALL", "1")
        b = _render(hip_lib, make(), w, h, True, True, passes, rpp, bounces)
        monkeypatch.delenv("RT_SAH_RECHECK_ALL")
        c = _render(hip_lib, make(), w, h, True, True, passes, rpp, bounces)
        assert b[3]["tri_tests"] > c[3]["tri_tests"] and b[3]["box_tests"] > c[3]["box_tests"]   # the switch did switch
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
        assert a[3]["rays"] == b[3]["rays"] and a[3]["shaded_hits"] == b[3]["shaded_hits"]


@pytest.mark.gpu
def test_root_table_octant_variants_same_image(hip_lib, oracle_lib):
    """Small SAH scenes keep their root table once per direction octant, every box in (near, far) order per axis, so that the
    root loop needs no min / max (rt_dev_trace.h::box_test_oct; the ROCT instantiations of k_trace).  The entry / exit values are
    the same numbers, so the image must be the one without the variants (RT_ROOT_OCT=0) - and the oracle's - bit for bit, in
    every culling mode; info() says which table the launch read."""
    w, h = 320, 180
    sc = build_suzanne(0)
    o = oracle_lib.create()
    o.load_scene(sc)
    o.load_camera(sc.cam)
    o.resize(w, h)
    o.render(1, 3, 1, 2, 6, 0)
    want = o.read_mean().view(np.uint32)
    old = os.environ.get("RT_ROOT_OCT")
    try:
        for cull in (0, 1, 2):
            got = {}
            for flag in ("0", "1"):
                os.environ["RT_ROOT_OCT"] = flag             # read by rt_create
                c = _ctx(hip_lib, build_suzanne(0), w, h, True, cull)
                c.render(1, 3, 1, 2, 6, 0)
                got[flag] = c.read_mean().view(np.uint32)
                assert c.info()["root_table_variants"] == (8 if flag == "1" else 1), c.info()
                c.close()
            assert np.array_equal(got["0"], got["1"]), f"cull mode {cull}: the octant table changes the image"
            assert np.array_equal(got["1"], want), f"cull mode {cull}: differs from the oracle"
    finally:
        if old is None:
            os.environ.pop("RT_ROOT_OCT", None)
        else:
            os.environ["RT_ROOT_OCT"] = old
    # a scene with analytic spheres, or on the reference's structure, has no such instantiation: the plain table
    c = _ctx(hip_lib, random_scene(3), 64, 36, True, 1)
    c.render(1, 1, 1, 1, 2, 0)
    assert c.info()["root_table_variants"] in (0, 1)
    c.close()
    c = _ctx(hip_lib, build_suzanne(0), 64, 36, False, 1)
    c.render(1, 1, 1, 1, 2, 0)
    assert c.info()["root_table_variants"] == 1
    c.close()


@pytest.mark.gpu
def test_vote_at_the_extremes_of_the_refill_threshold(hip_lib, oracle_lib):
    """The traversal loop has ONE way out: fewer than refill_below lanes traverse and some lane waits for the path block (round 3: the
    vote written for the scalar unit).  At the ends of the range - 1: a wave shades only when nobody traverses; 64: as soon as one lane
    has finished - and for every burst length the image must still be the oracle's, in the exact mode and with both structures culled."""
    w, h = 160, 90
    sc = build_suzanne(0)
    o = oracle_lib.create()
    o.load_scene(sc)
    o.load_camera(sc.cam)
    o.resize(w, h)
    o.render(1, 3, 1, 2, 5, 0)
    want = o.read_mean().view(np.uint32)
    old = {k: os.environ.get(k) for k in ("RT_REFILL_BELOW", "RT_BURST8")}
    try:
        for refill, burst in (("1", "3"), ("64", "3"), ("2", "0"), ("63", "8"), ("22", "64")):
            os.environ["RT_REFILL_BELOW"], os.environ["RT_BURST8"] = refill, burst      # read by rt_create
            for sah, cull in ((False, 0), (False, 1), (True, 1)):
                c = _ctx(hip_lib, build_suzanne(0), w, h, sah, cull)
                c.render(1, 3, 1, 2, 5, 0)
                got = c.read_mean().view(np.uint32)
                c.close()
                assert np.array_equal(got, want), f"refill {refill}, burst {burst}, sah {sah}, cull {cull}"
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
