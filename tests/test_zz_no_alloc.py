"""rt_render_static (include/rt_hip.h): the entry point that allocates nothing - its launches keep
the work counter, the deep node-stack entries and the per-(pass, pixel) radiance in the library's
file-scope device array.  Checked through the C-ABI against rt_render, bit for bit, on two calls in
a row (the second with a smaller input: the array keeps the first call's contents), in running-mean
and strided-sum mode, on the reference structure and on the SAH structure with culling, on a scene
whose node stack exceeds LDS (deep entries in the array), at 3840x2160 and with more passes than one launch holds.
Round 3: the property itself (free device memory and the event-timed kernel_ms do not move), the launches whose grid
the scratch caps, the state rule (pending pose updates are refused, rt_prepare flushes them) and the caller it is for:
the call captured into a hipGraph in global capture mode and replayed.  (The two RT_E_LIMIT returns of the static path -
BVH too deep for 16 MiB, a job larger than 48 MiB - cannot be reached through the C-ABI: RT_MAX_STACK bounds the depth
at 32 and a launch holds at most 1024 passes; they guard the constants, not an input.)

Collected last on purpose (test_zz_*): the round's end runs `pytest -x`."""
import ctypes

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


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    hip.hipStreamBeginCapture.argtypes = [ctypes.c_void_p, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    hip.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    hip.hipGraphExecDestroy.argtypes = [ctypes.c_void_p]
    hip.hipGraphDestroy.argtypes = [ctypes.c_void_p]
    return hip


def _free_bytes(hip):
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_static_allocates_nothing_and_creates_no_events(hip_lib):
    # the property the entry point exists for: free device memory and the event-timed kernel_ms do not move
    hip = _hip()
    sc = build_suzanne(0)
    ctx = make_ctx(hip_lib, sc, 640, 360)
    ctx.set_cull(True)
    ctx.render(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)          # everything rt_render allocates exists now
    ctx.sync()
    ms0, free0 = ctx.stats()["kernel_ms"], _free_bytes(hip)
    ctx.reset_accum()
    ctx.render_static(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    ctx.sync()
    assert _free_bytes(hip) == free0
    st = ctx.stats()
    assert st["kernel_ms"] == ms0 and st["launches"] == 2
    # a larger image than any rt_render of this context has seen: rt_render would grow its staging buffer here
    ctx.resize(1920, 1080)                                # (allocates the accumulator: allowed, it is not rt_render_static)
    free1 = _free_bytes(hip)
    ctx.render_static(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    ctx.sync()
    assert _free_bytes(hip) == free1
    ctx.close()


def test_static_refuses_pending_pose_updates_and_prepare_flushes_them(hip_lib):
    sc = build_suzanne(0)
    ctx = make_ctx(hip_lib, sc, 160, 90)
    ctx.render(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    a = ctx.read_accum()[0].copy()
    m = sc.meshes[0]
    ctx.update_mesh_pose(m.id, m.csys.pos, m.csys.quat)   # the normal animation order: pose, then render
    with pytest.raises(ffi.RtError) as e:
        ctx.render_static(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    assert e.value.code == -2 and "rt_prepare" in str(e.value)
    ctx.prepare()
    ctx.reset_accum()
    ctx.render_static(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    assert np.array_equal(_bits(a), _bits(ctx.read_accum()[0]))
    # materials alone pending: one more kernel launch, no copy from pageable memory, no synchronisation - goes through
    ctx.update_scene(sc)                                  # no mesh flagged: rt_set_materials only
    ctx.reset_accum()
    ctx.render_static(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    assert np.array_equal(_bits(a), _bits(ctx.read_accum()[0]))
    ctx.close()


@pytest.mark.parametrize("sah,cull", [(False, 0), (True, 1)])
def test_static_under_stream_capture(hip_lib, sah, cull):
    """The caller the entry point is for: rt_render_static captured into a hipGraph (global capture mode, in which any
    hipMalloc / hipFree / synchronisation on the thread invalidates the capture), replayed twice."""
    hip = _hip()
    sc = build_suzanne(0)
    ctx = hip_lib.create(0)
    ctx.set_accel(sah)
    ctx.load_scene(sc)
    ctx.load_camera(sc.cam)
    ctx.resize(320, 180)
    ctx.set_cull(cull)
    ctx.render(1, 3, 1, 1, 4, ffi.RT_ACCUM_MEAN)          # flushes the poses; the image to match
    a = ctx.read_accum()[0].copy()
    ctx.reset_accum()
    ctx.render_static(1, 3, 1, 1, 4, ffi.RT_ACCUM_MEAN)   # first call of this instantiation: occupancy query outside the capture
    ctx.sync()
    stream = ctypes.c_void_p(ctx.stream())
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0      # hipStreamCaptureModeGlobal
    try:
        ctx.render_static(1, 3, 1, 1, 4, ffi.RT_ACCUM_MEAN)
    finally:
        rc = hip.hipStreamEndCapture(stream, ctypes.byref(graph))
    assert rc == 0 and graph.value, f"capture was invalidated: hipStreamEndCapture -> {rc}"
    assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
    for _ in range(2):
        ctx.reset_accum()
        assert hip.hipGraphLaunch(gexec, stream) == 0
        ctx.sync()
        assert np.array_equal(_bits(a), _bits(ctx.read_accum()[0]))
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    ctx.close()


def test_static_grid_capped_by_the_scratch(hip_lib, monkeypatch):
    # (a) staging: 600 passes in one launch = jobs of 600 samples; 48 MiB hold the two slots of 873 workgroups' waves, fewer than
    # the chip takes; (b) deep entries: RT_STACK_LDS=2 at 1080p puts ~13 entries per lane of a full grid into 16 MiB
    sc = build_suzanne(0)
    ctx = make_ctx(hip_lib, sc, 160, 90)
    out = []
    for fn in (ctx.render, ctx.render_static):
        ctx.reset_accum()
        fn(1, 600, 1, 1, 2, ffi.RT_ACCUM_MEAN)
        out.append(ctx.read_accum()[0].copy())
    assert out[0].max() > 0 and np.array_equal(_bits(out[0]), _bits(out[1]))
    ctx.close()
    monkeypatch.setenv("RT_STACK_LDS", "2")
    ctx = make_ctx(hip_lib, build_suzanne(0), 1920, 1080)
    out = []
    for fn in (ctx.render, ctx.render_static):
        ctx.reset_accum()
        fn(1, 2, 1, 1, 4, ffi.RT_ACCUM_MEAN)
        out.append(ctx.read_accum()[0].copy())
    info = ctx.info()
    deep = info["stack_entries"] - info["stack_entries_lds"]
    assert deep > 0 and (16 << 20) // (deep * 4 * 256) < info["workgroups_per_cu"] * 256, info      # the cap is what sized the grid
    assert np.array_equal(_bits(out[0]), _bits(out[1]))
    ctx.close()
