"""BASELINE.json configs[3]: Suzanne midpoint-subdivided (5 levels = 991 232 triangles) in the
room, BVH built natively by rt_bvh_build.  The scene no longer fits one XCD's L2, the BVH hits
the depth-20 stop (leaves of up to 278 triangles), and the pose/refit kernels run at scale."""
import numpy as np
import pytest

from conftest import make_ctx
from raytracer_amd.scripts.scenes import final_scene_numba as S
from raytracer_amd.scripts.scenes.chunker import NODE_DTYPE


def _scene(levels, lib, frame=0):
    sc = S.build(subdivide=levels, lib=lib)
    sc.animate_frame(frame)
    return sc


def _run(lib, sc, w, h, cull, passes, bounces=4):
    c = make_ctx(lib, sc, w, h)
    c.set_cull(cull)
    c.enable_stats(True)
    ids, ts = c.first_hit(1)
    c.render(1, passes, 1, 1, bounces, 0)
    return ids, ts, c.read_mean(), c.stats(), c


def test_subdivision_counts(hip_lib_cpu):
    sc = _scene(2, hip_lib_cpu)
    assert sc.count_triangles() == 968 * 16 + 84
    p = sc.meshes[0].triangles.pos0
    # children tile their parent: same total area (to rounding), same bounding box
    p1 = S.build().meshes[0].triangles.pos0

    def area(q):
        return 0.5 * np.linalg.norm(np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]).astype(np.float64), axis=1).sum()

    assert abs(area(p) - area(p1)) < 1e-4 * area(p1)
    assert np.array_equal(p.reshape(-1, 3).min(0), p1.reshape(-1, 3).min(0))
    assert np.array_equal(p.reshape(-1, 3).max(0), p1.reshape(-1, 3).max(0))


@pytest.mark.gpu
def test_level2_bit_exact(oracle_lib, hip_lib):
    io, to, a, so, _ = _run(oracle_lib, _scene(2, oracle_lib), 128, 72, False, 2)
    ig, tg, b, sg, _ = _run(hip_lib, _scene(2, hip_lib), 128, 72, False, 2)
    assert np.array_equal(io, ig) and np.array_equal(to.view(np.uint32), tg.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert so["rays"] == sg["rays"] and so["tri_tests"] == sg["tri_tests"]
    ic, tc, c, _, _ = _run(hip_lib, _scene(2, hip_lib), 128, 72, True, 2)
    assert np.array_equal(ig, ic) and np.array_equal(b.view(np.uint32), c.view(np.uint32))


@pytest.mark.gpu
def test_million_triangles(oracle_lib, hip_lib):
    # frame 7: the subject is posed (spin + bob).  A scene is loaded before the next one is built
    # (building resets the global triangle / graph registries, like the reference's modules).
    io, to, a, so, co = _run(oracle_lib, _scene(5, oracle_lib, frame=7), 96, 54, False, 1)
    sc_g = _scene(5, hip_lib, frame=7)
    assert sc_g.count_triangles() == 991316
    ig, tg, b, sg, cg = _run(hip_lib, sc_g, 96, 54, False, 1)
    # device pose + level-by-level refit of 381 k nodes == oracle
    assert np.array_equal(co.read_tris28().view(np.uint32), cg.read_tris28().view(np.uint32))
    assert co.read_nodes48(NODE_DTYPE).tobytes() == cg.read_nodes48(NODE_DTYPE).tobytes()
    assert np.array_equal(io, ig) and np.array_equal(to.view(np.uint32), tg.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert so["rays"] == sg["rays"] and so["tri_tests"] == sg["tri_tests"]
    # full-resolution property: culled traversal == strict traversal, 4K first hits + 1080p radiance
    s = make_ctx(hip_lib, sc_g, 3840, 2160)
    i0, t0 = s.first_hit(1)
    s.set_cull(True)
    i1, t1 = s.first_hit(1)
    assert np.array_equal(i0, i1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert len(np.unique(i0)) > 100000                             # most of the visible micro-triangles are distinct
    s.resize(960, 540)
    s.render(1, 2, 1, 1, 4, 0)
    m1 = s.read_mean()
    s.set_cull(False)
    s.reset_accum()
    s.render(1, 2, 1, 1, 4, 0)
    assert np.array_equal(m1.view(np.uint32), s.read_mean().view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_million_triangles_sah_culled_deep_vs_oracle(oracle_lib, hip_lib, mode):
    """The mode `bench.py --workload subdiv4k` times: 991 316 triangles on the SAH structure with distance culling, node
    stack deeper than its LDS part (DEEP instantiation, the scheduler's refill_below 28 branch for scenes that do not fit
    the L2) - against the ORACLE (raytracer.fs.glsl:518-641 restated), not only against the exact mode.  mode 2 = the
    distance band alone (rt_set_cull(ctx, RT_CULL_BAND))."""
    sc = _scene(5, hip_lib, frame=7)
    W, H = 480, 270
    co = make_ctx(oracle_lib, sc, W, H)
    io, to = co.first_hit(1)
    co.render(1, 2, 1, 1, 4, 0)
    a = co.read_mean()
    co.close()
    cg = hip_lib.create()
    cg.set_accel(True)
    cg.load_scene(sc)
    cg.load_camera(sc.cam)
    cg.resize(W, H)
    cg.set_cull(mode)
    ig, tg = cg.first_hit(1)
    cg.render(1, 2, 1, 1, 4, 0)
    b = cg.read_mean()
    info = cg.info()
    cg.close()
    assert info["accel"] == 1 and info["n_slots"] > 100000
    assert info["stack_entries_lds"] < info["stack_entries"], info           # DEEP: entries beyond LDS go to global memory
    assert a.max() > 0 and len(np.unique(io)) > 10000
    assert np.array_equal(io, ig), f"{(io != ig).sum()} first-hit ids differ"
    assert np.array_equal(to.view(np.uint32), tg.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).any(axis=2).sum()} pixels differ"
