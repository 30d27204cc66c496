"""The oracle's variant switches (oracle/rt_oracle.c, rto_set_variant) and its epsilon-grazing mask
(SURVEY.md 9.7).  The default oracle is ONE admissible reading of the shader's text; the variants
are the others: true division in the slab test (raytracer.fs.glsl:425-442 as written), the C
library's sin/cos/acos/exp/log where the GL driver's would run, the `doHitBackside` carry-over of
:522, 605-610.  Here (CPU, small sizes): the switches are live, do what they say and nothing else.
The distances at 1920x1080 are measured on the GPU box (tests/test_gpu_envelope.py)."""
import ctypes as C

import numpy as np
import pytest

import synth
from conftest import build_suzanne, make_ctx

W, H = 160, 90


def variant_ctx(lib, sc, w, h, true_div=0, libm=0, carry=0):
    c = make_ctx(lib, sc, w, h)
    lib.dll.rto_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    assert lib.dll.rto_set_variant(c._h, true_div, libm, carry) == 0
    return c


def grazing(lib, ctx, pass_index=1, only=None):
    lib.dll.rto_first_hit_grazing.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    m = np.zeros((ctx.height, ctx.width), dtype=np.uint8)
    o = None if only is None else np.ascontiguousarray(only, dtype=np.uint8)
    rc = lib.dll.rto_first_hit_grazing(ctx._h, pass_index, m.ctypes.data, None if o is None else o.ctypes.data)
    assert rc == 0
    return m


def _render(c, passes=3, bounces=4):
    ids, ts = c.first_hit(1)
    c.render(1, passes, 1, 1, bounces, 0)
    return ids, ts, c.read_mean()


def test_libm_variant_leaves_primary_rays_alone_and_moves_radiance_by_rounding(oracle_lib):
    base = _render(variant_ctx(oracle_lib, build_suzanne(0), W, H))
    libm = _render(variant_ctx(oracle_lib, build_suzanne(0), W, H, libm=1))
    # primary rays use + - * / sqrt only: identical ids and distances
    assert np.array_equal(base[0], libm[0]) and np.array_equal(base[1].view(np.uint32), libm[1].view(np.uint32))
    # radiance in this scene is a product of material constants along the path (no cosine factor, no
    # transmission): it is piecewise constant in the bounce directions, so a last-bit change of sin/cos/acos
    # moves a pixel only when it flips WHICH surface a bounce ray meets
    a, b = base[2], libm[2]
    tol = 1e-3 * np.maximum(np.abs(a), 1e-3 * a.mean())
    within = (np.abs(a - b) <= tol).all(axis=2)
    print(f"libm variant, Suzanne room: {(a != b).any(axis=2).sum()} of {W*H} pixels differ in some bit, {(~within).sum()} beyond 1e-3")
    assert within.mean() > 0.99
    # the glass of the stress scene attenuates by exp(-a * distance) (:822-847): continuous in the directions,
    # so there the two sets of functions cannot agree bit for bit - the switch is live
    sa = _render(variant_ctx(oracle_lib, synth.stress_scene(12, spheres=False), W, H), passes=2, bounces=8)
    sb = _render(variant_ctx(oracle_lib, synth.stress_scene(12, spheres=False), W, H, libm=1), passes=2, bounces=8)
    differ = (sa[2] != sb[2]).any(axis=2)
    assert differ.any(), "the switch is dead"
    tol = 1e-3 * np.maximum(np.abs(sa[2]), 1e-3 * sa[2].mean())
    within = (np.abs(sa[2] - sb[2]) <= tol).all(axis=2)
    print(f"libm variant, glass scene: {differ.sum()} of {W*H} pixels differ in some bit, {(~within).sum()} beyond 1e-3")
    assert within.mean() > 0.9


def test_true_division_variant_matches_the_env_switch(oracle_lib, monkeypatch):
    a = _render(variant_ctx(oracle_lib, build_suzanne(0), W, H, true_div=1), passes=1)
    monkeypatch.setenv("RT_ORACLE_TRUE_DIV", "1")
    b = _render(make_ctx(oracle_lib, build_suzanne(0), W, H), passes=1)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_carry_over_variant_only_matters_inside_solids(oracle_lib):
    # the Suzanne room has no transmissive material: a ray is never in a solid, the carry-over is unreachable
    a = _render(variant_ctx(oracle_lib, build_suzanne(0), W, H))
    b = _render(variant_ctx(oracle_lib, build_suzanne(0), W, H, carry=1))
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    # the stress scene has glass: rays travel inside meshes; the carry-over may let a ray inside one mesh hit
    # the BACK faces of other meshes that are listed after a triangle of its own mesh
    sa = _render(variant_ctx(oracle_lib, synth.stress_scene(12, spheres=False), W, H), passes=4, bounces=8)
    sb = _render(variant_ctx(oracle_lib, synth.stress_scene(12, spheres=False), W, H, carry=1), passes=4, bounces=8)
    assert np.array_equal(sa[0], sb[0])                      # primary rays are outside every solid
    n = int((sa[2] != sb[2]).any(axis=2).sum())
    print(f"carry-over variant, glass scene: {n} of {W*H} pixels differ after 4 passes x 8 bounces")
    assert n < 0.05 * W * H


def test_grazing_mask_known_answers(oracle_lib):
    """Two coplanar triangles sharing the edge x = 0, camera looking straight at it: the pixel column
    whose ray meets the shared edge is grazing (edge value 0 and a second hit at the same distance)."""
    from raytracer_amd.scripts.classes import Material, Scene
    from raytracer_amd.scripts.scenes import chunker
    from raytracer_amd.scripts.scenes.utils import reset_global_datas
    from raytracer_amd.vecmath import Vector4

    reset_global_datas()
    sc = Scene()
    mat = Material(Vector4((0.8, 0.8, 0.8, 1.0), dtype="f4"))
    n = np.tile(np.array([0, 0, -1], np.float32), (3, 1))
    left = np.array([[[-1, -1, 5], [0, 1, 5], [0, -1, 5]]], np.float32)      # both wound so that N faces the camera (-z)
    right = np.array([[[0, -1, 5], [0, 1, 5], [1, -1, 5]]], np.float32)
    for p in (left, right):
        if np.cross(p[0, 1] - p[0, 0], p[0, 2] - p[0, 0])[2] > 0:
            p[0] = p[0, [0, 2, 1]]
        synth.add_mesh(sc, p, n[None], mat)
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    cam = sc.cam
    cam.csys.set_pos([0.0, 0.0, 0.0])
    cam.fov = 30
    cam.near_plane = 1.0
    cam.antialias_strength = 0.0
    cam.depth_of_field_strength = 0.0
    w, h = 65, 37                                            # odd width: the centre column looks along x = 0 exactly
    c = make_ctx(oracle_lib, sc, w, h)
    ids, ts = c.first_hit(1)
    m = grazing(oracle_lib, c)
    mid = w // 2
    hit_rows = np.where(ids[:, mid] >= 0)[0]
    assert len(hit_rows) > 5
    assert (m[hit_rows, mid] & 3 == 3).all(), "shared edge: second candidate at the same t and a zero edge value"
    inner = m[h // 2, mid - 4:mid - 1]
    assert (ids[h // 2, mid - 4:mid - 1] == 0).all() and (inner == 0).all()   # well inside the left triangle: not grazing
    # `only`: pixels outside it are not looked at
    only = np.zeros((h, w), np.uint8)
    only[h // 2, mid] = 1
    m2 = grazing(oracle_lib, c, only=only)
    assert m2[h // 2, mid] == m[h // 2, mid] and m2.sum() == m2[h // 2, mid]
