"""rt_oracle.c against the independent scalar-numpy restatement oracle/np_ref.py: bit-for-bit.
Two separately written restatements agreeing is what pins the oracle in the absence of any
reference fixture."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, build_suzanne, make_ctx

sys.path.insert(0, str(ROOT / "oracle"))
import np_ref  # noqa: E402

F = np.float32


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_det_math_bit_exact(oracle_lib):
    d = oracle_lib.dll
    d.rto_det_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(1)
    cases = {
        0: rng.uniform(0, 6.3, 300), 1: rng.uniform(0, 6.3, 300), 2: rng.uniform(-1, 1, 300),
        3: np.concatenate([rng.uniform(-104, 5, 300), [-200.0, 0.0, -np.inf]]),
        4: np.concatenate([np.exp(rng.uniform(-40, 40, 300)), [0.0, 1.0, 1e-40, np.inf]]),
    }
    for which, xs in cases.items():
        xs = xs.astype(F)
        out = np.empty_like(xs)
        d.rto_det_math(which, _fp(xs), _fp(out), xs.size)
        for x, got in zip(xs, out):
            if which == 0:
                exp = np_ref.det_sincos(x)[0]
            elif which == 1:
                exp = np_ref.det_sincos(x)[1]
            else:
                exp = (np_ref.det_acos, np_ref.det_exp, np_ref.det_log)[which - 2](x)
            assert F(exp).view(np.uint32) == F(got).view(np.uint32), (which, x, exp, got)


@pytest.fixture(scope="module")
def posed(oracle_lib):
    """Oracle context + the posed std140 buffers it works on, for np_ref."""
    from raytracer_amd.scripts import functions as Fn
    from raytracer_amd.scripts.scenes.chunker import NODE_DTYPE

    sc = build_suzanne(0)
    W, H = 48, 27
    ctx = make_ctx(oracle_lib, sc, W, H)
    bufs = Fn.export_scene_buffers(sc, rest_pose=True)
    bufs["tris28"] = ctx.read_tris28()
    bufs["nodes48"] = ctx.read_nodes48(NODE_DTYPE)
    cam = Fn.camera_uniforms(sc.cam)
    return ctx, np_ref.Scene(bufs), cam, W, H


def test_primary_rays_bit_exact(oracle_lib, posed):
    ctx, _, cam, W, H = posed
    d = oracle_lib.dll
    d.rto_primary_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(2)
    for _ in range(25):
        px, py, p = int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(1, 1000))
        out = np.zeros(6, dtype=F)
        st = C.c_uint32(0)
        assert d.rto_primary_ray(ctx._h, px, py, p, _fp(out), C.byref(st)) == 0
        o, dr, r = np_ref.primary_ray(cam, W, H, px, py, p)
        assert np.array_equal(out[:3].view(np.uint32), o.view(np.uint32))
        assert np.array_equal(out[3:].view(np.uint32), dr.view(np.uint32))
        assert st.value == r.s


def test_first_hit_matches_npref(posed):
    ctx, scn, cam, W, H = posed
    ids, ts = ctx.first_hit(1)
    rng = np.random.default_rng(3)
    for _ in range(30):
        px, py = int(rng.integers(W)), int(rng.integers(H))
        o, d, _r = np_ref.primary_ray(cam, W, H, px, py, 1)
        h = np_ref.closest_hit(scn, o, d, -1, -1, False)
        if h is None:
            assert ids[py, px] == -1
        else:
            assert ids[py, px] == h["facetId"]
            assert F(ts[py, px]).view(np.uint32) == F(h["dst"]).view(np.uint32)


def test_pixel_radiance_bit_exact(oracle_lib, posed):
    ctx, scn, cam, W, H = posed
    d = oracle_lib.dll
    d.rto_pixel_radiance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(4)
    nonzero = 0
    for k in range(10):
        px, py, p = int(rng.integers(W)), int(rng.integers(H)), int(rng.integers(1, 64))
        rpp, bounces = (2, 3) if k % 3 == 0 else (1, 4)
        out = np.zeros(3, dtype=F)
        assert d.rto_pixel_radiance(ctx._h, px, py, p, rpp, bounces, _fp(out)) == 0
        exp = np_ref.pixel_radiance(scn, cam, W, H, px, py, p, rpp, bounces)
        assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), (px, py, p, out, exp)
        nonzero += bool(out.any())
    assert nonzero >= 1
