"""The reference's animated 192-mesh scene (scripts/scenes/dragon_scene_animated.py): 188 cubes
falling into place over 120 frames, camera from the dragon keyframe CSV.  CPU: the recipe and the
clock (hand-derived known answers).  GPU: bit-exact against the oracle across pose updates of 188
meshes (device pose + refit of the mesh trees AND of the top-level tree over the 192 roots), and
culling / the SAH structure give the same image."""
import numpy as np
import pytest

from conftest import make_ctx
from raytracer_amd.scripts.scenes import dragon_scene_animated as A

W, H = 128, 72


@pytest.fixture(scope="module")
def scene():
    return A.build()


def test_clock_known_answers():
    # frames 0..30 advance by exactly 1/60 s each: the trapezoid sum is i/60
    assert A.get_wall_clock_time(0) == 0.0
    assert abs(A.get_wall_clock_time(30) - 0.5) < 1e-12
    assert abs(A.get_wall_clock_dt(45) - (0.5 * 1 + 0.5 * 0.001) / 60) < 1e-15      # smoothstep(30,60,45) = 1/2
    assert A.get_wall_clock_dt(60) == pytest.approx(0.001 / 60) and A.get_wall_clock_dt(119) < 1e-7
    c = np.array([A.get_wall_clock_time(i) for i in range(A.N_FRAMES)])
    assert (np.diff(c) >= 0).all() and c[60] > 0.7 and c[119] - c[60] < 0.001   # slow motion after frame 60
    # the cubes end where the files put them and start 10 * clock(119) higher (+y)
    assert np.array_equal(A.cube_offset(119), np.zeros(3, np.float32))
    o0 = A.cube_offset(0)
    assert o0[0] == 0 and o0[2] == 0 and o0[1] == np.float32(10.0 * c[119]) and 7.0 < o0[1] < 9.0


def test_recipe(scene):
    sc = scene
    assert len(sc.meshes) == 192 and sc.count_triangles() == 12658 and len(sc.cube_meshes) == 188
    assert all(m.csys is sc.meshes[0].csys for m in sc.meshes)          # one shared pose object
    assert sc.keyframes.shape == (120, 7)
    assert (sc.cam.fov, sc.cam.bounces_per_ray, sc.cam.rays_per_pixel, sc.cam.passes_per_frame) == (25, 3, 2, 20)
    assert sc.cam.near_plane == 9.4 and sc.cam.depth_of_field_strength == 0.002
    assert A.get_frame_number(sc, 57) == 57 and A.get_frame_number(sc, 120) == 0
    for m in sc.meshes:
        m.unflag_for_mesh_update()
    sc.animate_frame(10)
    assert [m.is_awaiting_mesh_update for m in sc.meshes[:4]] == [False] * 4      # dragon, light, floor, base stay
    assert all(m.is_awaiting_mesh_update for m in sc.cube_meshes)
    assert np.array_equal(sc.meshes[0].csys.pos, A.cube_offset(10))               # ... though the shared pose moved
    sc.meshes[0].csys.pos = np.zeros(3, dtype=np.float32)


def _frames(ctxs, sc, frames, passes=2):
    out = []
    for f in frames:
        sc.animate_frame(f)
        imgs = []
        for c in ctxs:
            for m in sc.cube_meshes:              # the flags are cleared by each context's update
                m.flag_for_mesh_update()
            c.update_scene(sc)
            c.load_camera(sc.cam)
            c.reset_accum()
            ids, ts = c.first_hit(1)
            c.render(1, passes, 1, sc.cam.rays_per_pixel, sc.cam.bounces_per_ray, 0)
            imgs.append((ids, ts, c.read_mean()))
        out.append(imgs)
    return out


@pytest.mark.gpu
def test_animated_dragon_bit_exact(oracle_lib, hip_lib, scene):
    sc = scene
    sc.meshes[0].csys.pos = np.zeros(3, dtype=np.float32)           # uploaded at the rest pose, as the reference does
    ref = make_ctx(oracle_lib, sc, W, H)
    strict = make_ctx(hip_lib, sc, W, H)
    cull = make_ctx(hip_lib, sc, W, H)
    cull.set_cull(True)
    sah = hip_lib.create()
    sah.set_accel(True)
    sah.load_scene(sc)
    sah.load_camera(sc.cam)
    sah.resize(W, H)
    sah.set_cull(True)
    frames = (0, 40, 70, 119)
    res = _frames([ref, strict, cull, sah], sc, frames)
    moved = 0
    for f, imgs in zip(frames, res):
        (io, to, a), rest = imgs[0], imgs[1:]
        for name, (ig, tg, b) in zip(("strict", "culled", "SAH"), rest):
            assert np.array_equal(io, ig), f"frame {f} {name}: first hits"
            assert np.array_equal(to.view(np.uint32), tg.view(np.uint32)), f"frame {f} {name}: distances"
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f} {name}: {(a != b).any(axis=2).sum()} pixels differ"
        moved += int((io != res[-1][0][0]).sum())
    assert moved > 100                                               # the cubes did move through the view
    for ctx in (strict, cull, sah):
        i = ctx.info()
        # the top-level tree follows the poses; a 29-level stack keeps 6 workgroups per CU by capping its LDS part
        assert i["top_level_tree"] == 1 and i["top_level_rebuilds"] >= len(frames)
        assert i["stack_entries_lds"] < i["stack_entries"] and i["workgroups_per_cu"] >= 6
    assert len(np.unique(res[-1][0][0])) > 300
