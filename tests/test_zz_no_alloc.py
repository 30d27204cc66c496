"""rt_render_static (include/rt_hip.h): the entry point that allocates nothing - its launches keep
the work counter, the deep node-stack entries and the per-(pass, pixel) radiance in the library's
file-scope device array.  Checked through the C-ABI against rt_render, bit for bit, on two calls in
a row (the second with a smaller input: the array keeps the first call's contents), in running-mean
and strided-sum mode, on the reference structure and on the SAH structure with culling, on a scene
whose node stack exceeds LDS (deep entries in the array), at 3840x2160 and with more passes than one launch holds.

Collected last on purpose (test_zz_*): the round's end runs `pytest -x`."""
import numpy as np
import pytest

from conftest import build_suzanne, make_ctx
from raytracer_amd import ffi

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _both(ctx, w, h, first, n, stride, rpp, bounces, flags):
    ctx.resize(w, h)
    out = []
    for fn in (ctx.render, ctx.render_static):
        ctx.reset_accum()
        fn(first, n, stride, rpp, bounces, flags)
        out.append(ctx.read_accum()[0].copy())
    return out


@pytest.mark.parametrize("sah,cull", [(False, False), (True, True)])
def test_static_equals_render_two_calls_in_a_row(hip_lib, sah, cull):
    sc = build_suzanne(0)
    ctx = hip_lib.create(0)
    ctx.set_accel(sah)
    ctx.load_scene(sc)
    ctx.load_camera(sc.cam)
    ctx.set_cull(cull)
    # first call: 480x270, 5 passes, running mean
    a, b = _both(ctx, 480, 270, 1, 5, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    assert a.max() > 0
    assert np.array_equal(_bits(a), _bits(b)), f"{(a != b).any(axis=2).sum()} pixels differ"
    # second call, smaller input, other mode: 160x90, 3 strided passes x 2 rays, SUM
    a2, b2 = _both(ctx, 160, 90, 2, 3, 4, 2, 8, ffi.RT_ACCUM_SUM)
    assert a2.max() > 0
    assert np.array_equal(_bits(a2), _bits(b2)), f"{(a2 != b2).any(axis=2).sum()} pixels differ"
    # and the first size again, straight after the small one
    a3, b3 = _both(ctx, 480, 270, 1, 5, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    assert np.array_equal(_bits(a3), _bits(a)) and np.array_equal(_bits(b3), _bits(a))
    ctx.close()


def test_static_full_size_frame_in_one_launch(hip_lib):
    # 1920x1080 x 5 passes: the staging slots of all waves fit the array (jobs of 256 samples), one launch like rt_render
    sc = build_suzanne(0)
    ctx = make_ctx(hip_lib, sc, 1920, 1080)
    ctx.set_cull(True)
    a, b = _both(ctx, 1920, 1080, 1, 5, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    assert np.array_equal(_bits(a), _bits(b))
    assert ctx.stats()["launches"] == 2
    ctx.close()


def test_static_deep_scene_uses_the_array_for_stack_entries(hip_lib, monkeypatch):
    # RT_STACK_LDS=2: nearly every deferred child of the Suzanne BVH goes to the deep-entry store
    monkeypatch.setenv("RT_STACK_LDS", "2")
    sc = build_suzanne(0)
    ctx = make_ctx(hip_lib, sc, 320, 180)
    a, b = _both(ctx, 320, 180, 1, 3, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    info = ctx.info()
    assert info["stack_entries_lds"] < info["stack_entries"]
    assert a.max() > 0 and np.array_equal(_bits(a), _bits(b))
    ctx.close()
    monkeypatch.delenv("RT_STACK_LDS")
    ref = make_ctx(hip_lib, build_suzanne(0), 320, 180)
    ref.render(1, 3, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    assert np.array_equal(_bits(ref.read_accum()[0]), _bits(a))
    ref.close()


def test_static_takes_any_image_size_and_many_passes(hip_lib):
    # no per-(pass, pixel) buffer any more: 3840x2160 works, and 1100 passes (more than one launch holds) at a small size
    sc = build_suzanne(0)
    ctx = make_ctx(hip_lib, sc, 3840, 2160)
    a, b = _both(ctx, 3840, 2160, 1, 1, 1, 1, 2, ffi.RT_ACCUM_MEAN)
    assert a.max() > 0 and np.array_equal(_bits(a), _bits(b))
    a, b = _both(ctx, 48, 27, 1, 1100, 1, 1, 1, ffi.RT_ACCUM_MEAN)
    assert a.max() > 0 and np.array_equal(_bits(a), _bits(b))
    ctx.close()
