"""Seeded random scenes (triangle soups, boxes, tessellated spheres, analytic spheres; random
materials incl. alpha, transmission, mirrors, emitters; 3 - 60 meshes so that both the LDS root
mask and the top-level tree are taken; random cameras with depth of field) and the
axis-aligned-ray corner of the slab test: HIP STRICT == oracle bit for bit, HIP CULL == HIP
STRICT; the SAH structure with culling (the bench mode) == oracle on all of them."""
import numpy as np
import pytest

from conftest import make_ctx
from raytracer_amd.scripts.classes import Material, Scene, Sphere
from raytracer_amd.scripts.numba_utils.classes import Csys
from raytracer_amd.scripts.scenes import chunker
from raytracer_amd.scripts.scenes.utils import reset_global_datas
from raytracer_amd.vecmath import Vector3, Vector4
from synth import add_mesh, box_triangles, uv_sphere

F = np.float32


def _material(r):
    kind = r.integers(0, 6)
    col = Vector4((*r.uniform(0.1, 1.0, 3), 1.0), dtype="f4")
    if kind == 0:
        return Material(Vector4((0, 0, 0, 1.0), dtype="f4"), emissionColor=Vector3(r.uniform(0.5, 1, 3), dtype="f4"),
                        emissionStrength=float(r.uniform(1, 8)))
    if kind == 1:
        return Material(col, smoothness=float(r.uniform(0.8, 1)), specularStrength=1.0)                  # mirror
    if kind == 2:
        return Material(col, smoothness=1.0, specularStrength=float(r.uniform(0, 0.2)), transmission=float(r.uniform(0.5, 1)),
                        ior=float(r.uniform(1.1, 1.8)), specularColor=Vector3(r.uniform(0.5, 1, 3), dtype="f4"))   # glass
    if kind == 3:
        col = Vector4((*r.uniform(0.1, 1.0, 3), float(r.uniform(0.1, 0.9))), dtype="f4")
        return Material(col, transparent_from_behind=bool(r.integers(0, 2)))                           # alpha veil
    return Material(col, smoothness=float(r.uniform(0, 1)), specularStrength=float(r.uniform(0, 0.7)), metallic=float(r.uniform(0, 1)))


def random_scene(seed):
    reset_global_datas()
    r = np.random.default_rng(seed)
    sc = Scene()
    sc.atmosphere_material = Material(Vector4((*r.uniform(0.7, 1, 3), 1.0), dtype="f4"), transmission=float(r.uniform(0.9, 1)), ior=1.0)
    n_meshes = int(r.choice([3, 9, 31, 32, 33, 60]))
    add_mesh(sc, *box_triangles((-8, -2.2, -8), (8, -2, 8)), _material(r))                              # something to land on
    for _ in range(n_meshes - 1):
        c = Csys()
        c.set_pos(r.uniform((-4, -1.5, -3), (4, 3, 5)))
        c.rxg(float(r.uniform(0, 360))).ryg(float(r.uniform(0, 360))).rzg(float(r.uniform(0, 360)))
        shape = r.integers(0, 3)
        if shape == 0:
            h = r.uniform(0.1, 0.8, 3)
            pos, nrm = box_triangles(-h, h)
        elif shape == 1:
            pos, nrm = uv_sphere((0, 0, 0), float(r.uniform(0.2, 0.9)), int(r.integers(5, 12)), int(r.integers(3, 8)))
        else:                                                                                         # soup incl. slivers
            n = int(r.integers(1, 40))
            ctr = r.uniform(-0.5, 0.5, (n, 1, 3))
            pos = (ctr + r.normal(0, 0.25, (n, 3, 3)) * r.choice([1.0, 1e-3], (n, 1, 1))).astype(F)
            nrm = r.normal(0, 1, (n, 3, 3)).astype(F)
        add_mesh(sc, pos, nrm, _material(r), c)
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    for _ in range(int(r.integers(0, 4))):
        sc.spheres.append(Sphere(Vector3(r.uniform((-4, -1, -2), (4, 3, 5))), float(r.uniform(0.2, 1.0)), _material(r)))
    cam = sc.cam
    cam.csys.set_pos(r.uniform((-2, 0, -10), (2, 3, -7)))
    cam.csys.ryg(float(r.uniform(-10, 10))).rxg(float(r.uniform(-10, 5)))
    cam.fov = float(r.uniform(25, 60))
    cam.near_plane = float(r.uniform(5, 10))
    cam.depth_of_field_strength = float(r.choice([0.0, 0.01, 0.05]))
    cam.antialias_strength = float(r.choice([0.0, 0.002]))
    return sc


def _render(lib, sc, w, h, cull, passes, rpp, bounces, sah=False):
    c = lib.create()
    c.set_accel(sah)                 # the oracle ignores it
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.resize(w, h)
    c.set_cull(cull)
    c.enable_stats(True)
    ids, ts = c.first_hit(1)
    c.render(1, passes, 1, rpp, bounces, 0)
    return ids, ts, c.read_mean(), c.stats()


def test_random_scene_is_renderable_by_the_oracle(oracle_lib):
    ids, ts, img, st = _render(oracle_lib, random_scene(5), 48, 27, False, 1, 1, 3)
    assert np.isfinite(img).all() and (ids >= 0).any() and st["rays"] > 48 * 27


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_scenes_bit_exact(oracle_lib, hip_lib, seed):
    w, h, passes, rpp, bounces = 96, 54, 2, 2, 6
    io, to, a, so = _render(oracle_lib, random_scene(seed), w, h, False, passes, rpp, bounces)
    ig, tg, b, sg = _render(hip_lib, random_scene(seed), w, h, False, passes, rpp, bounces)
    assert np.array_equal(io, ig), f"seed {seed}: {(io != ig).sum()} first-hit ids differ"
    assert np.array_equal(to.view(np.uint32), tg.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"seed {seed}: {(a != b).any(axis=2).sum()} pixels differ"
    assert so["rays"] == sg["rays"] and so["tri_tests"] == sg["tri_tests"] and so["shaded_hits"] == sg["shaded_hits"]
    ic, tc, c, _ = _render(hip_lib, random_scene(seed), w, h, True, passes, rpp, bounces)
    assert np.array_equal(ig, ic) and np.array_equal(tg.view(np.uint32), tc.view(np.uint32))
    assert np.array_equal(b.view(np.uint32), c.view(np.uint32)), f"seed {seed}: CULL differs from STRICT in {(b != c).any(axis=2).sum()} pixels"
    # the bench mode (SAH structure + culling) directly against the ORACLE, not against another HIP mode
    isa, tsa, d, _ = _render(hip_lib, random_scene(seed), w, h, True, passes, rpp, bounces, sah=True)
    assert np.array_equal(io, isa) and np.array_equal(to.view(np.uint32), tsa.view(np.uint32)), f"seed {seed}: SAH first hits differ from the oracle"
    assert np.array_equal(a.view(np.uint32), d.view(np.uint32)), f"seed {seed}: SAH + culled differs from the oracle in {(a != d).any(axis=2).sum()} pixels"


def _axis_scene():
    """Identity camera at the origin, odd image size, no jitter: the centre pixel's ray is exactly
    (0, 0, 1) -> 1/d = inf on two axes; boxes with faces in the planes x = 0 and y = 0 make
    (b - o) * (1/d) = 0 * inf = NaN.  The reference's compare ladder decides those cases."""
    reset_global_datas()
    sc = Scene()
    sc.atmosphere_material = Material(Vector4((1, 1, 1, 1.0), dtype="f4"), transmission=1.0, ior=1.0)
    grey = Material(Vector4((0.7, 0.7, 0.7, 1.0), dtype="f4"))
    light = Material(Vector4((0, 0, 0, 1.0), dtype="f4"), emissionColor=Vector3((1, 1, 1), dtype="f4"), emissionStrength=3.0)
    add_mesh(sc, *box_triangles((0, 0, 4), (1, 1, 5)), grey)          # corner on the axis: faces x = 0 and y = 0
    add_mesh(sc, *box_triangles((-1, -1, 6), (0, 0, 7)), grey)        # the opposite corner
    add_mesh(sc, *box_triangles((-3, -3, 9), (3, 3, 9.5)), light)     # back wall, emits
    add_mesh(sc, *uv_sphere((0, 0, 2.0), 0.3, 8, 4), Material(Vector4((1, 1, 1, 0.3), dtype="f4")))   # pole on the axis, mostly see-through
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    sc.cam.fov = 50
    sc.cam.near_plane = 1.0
    sc.cam.antialias_strength = 0.0
    sc.cam.depth_of_field_strength = 0.0
    return sc


@pytest.mark.gpu
def test_axis_aligned_rays_and_nan_slabs(oracle_lib, hip_lib):
    w, h = 65, 37
    io, to, a, so = _render(oracle_lib, _axis_scene(), w, h, False, 3, 1, 4)
    ig, tg, b, sg = _render(hip_lib, _axis_scene(), w, h, False, 3, 1, 4)
    assert np.array_equal(io, ig) and np.array_equal(to.view(np.uint32), tg.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert so["rays"] == sg["rays"] and so["tri_tests"] == sg["tri_tests"]
    ic, tc, c, _ = _render(hip_lib, _axis_scene(), w, h, True, 3, 1, 4)
    assert np.array_equal(ig, ic) and np.array_equal(b.view(np.uint32), c.view(np.uint32))
    isa, tsa, d, _ = _render(hip_lib, _axis_scene(), w, h, True, 3, 1, 4, sah=True)     # NaN slabs through the widened SAH tests
    assert np.array_equal(io, isa) and np.array_equal(a.view(np.uint32), d.view(np.uint32))


def _grid_scene(seed):
    """Boxes with every face on a 0.5 grid around the optical axis of an identity camera, mirrors and glass among
    them (a reflection off an axis-aligned face keeps a zero direction component, so bounce rays carry them too).
    Rendered one pixel wide or high, EVERY primary ray has 1/d = inf on an axis and passes exactly along, or a
    rounding error beside, box faces of both structures."""
    reset_global_datas()
    r = np.random.default_rng(1000 + seed)
    sc = Scene()
    sc.atmosphere_material = Material(Vector4((1, 1, 1, 1.0), dtype="f4"), transmission=1.0, ior=1.0)
    mats = [Material(Vector4((0.7, 0.7, 0.7, 1.0), dtype="f4")),
            Material(Vector4((0.9, 0.9, 0.9, 1.0), dtype="f4"), smoothness=1.0, specularStrength=1.0),
            Material(Vector4((1, 1, 1, 1.0), dtype="f4"), smoothness=1.0, specularStrength=0.1, transmission=0.9, ior=1.4),
            Material(Vector4((1, 1, 1, 0.4), dtype="f4")),
            Material(Vector4((0, 0, 0, 1.0), dtype="f4"), emissionColor=Vector3((1, 1, 1), dtype="f4"), emissionStrength=2.0)]
    add_mesh(sc, *box_triangles((-4, -4, 12), (4, 4, 12.5)), mats[4])
    for _ in range(int(r.integers(6, 20))):
        lo = r.integers((-4, -4, 4), (4, 4, 22), 3) * 0.5
        hi = lo + r.integers(1, 4, 3) * 0.5
        add_mesh(sc, *box_triangles(tuple(lo), tuple(hi)), mats[int(r.integers(0, 5))])
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    sc.cam.fov = float(r.choice([30, 50, 90]))
    sc.cam.near_plane = 1.0
    sc.cam.antialias_strength = 0.0
    sc.cam.depth_of_field_strength = 0.0
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)))
@pytest.mark.parametrize("size", [(1, 129), (129, 1), (33, 33)])
def test_zero_direction_components_in_every_mode(oracle_lib, hip_lib, seed, size):
    w, h = size
    io, to, a, so = _render(oracle_lib, _grid_scene(seed), w, h, False, 4, 2, 6)
    assert (io >= 0).any()
    for sah in (False, True):
        for cull in (False, True):
            ig, tg, b, sg = _render(hip_lib, _grid_scene(seed), w, h, cull, 4, 2, 6, sah=sah)
            what = f"seed {seed} {w}x{h} sah={sah} cull={cull}"
            assert np.array_equal(io, ig), f"{what}: {(io != ig).sum()} first-hit ids differ"
            assert np.array_equal(to.view(np.uint32), tg.view(np.uint32)), what
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{what}: {(a != b).any(axis=2).sum()} pixels differ"
            assert so["rays"] == sg["rays"], what


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(6)))
def test_near_axis_rays_culling_differs_only_where_the_audit_says(oracle_lib, hip_lib, seed):
    """Camera 1e-6 beside the face planes of the grid scene, yawed by 1e-6 degrees: |1/d| = 6e7 on one axis, slab values
    of +-60 for a box face a micrometre away, bounce origins a rounding error beside box edges.  The exact modes of
    both structures stay bit-equal to the oracle.  The culling modes may lose a winner only in a query the oracle's
    audit flags (tests/test_oracle_cull_audit.py), so: no flagged query -> bit-equal, else at most that many pixels."""
    import ctypes

    from conftest import ORACLE_SO
    from test_oracle_cull_audit import near_axis_scene

    audit = ctypes.CDLL(str(ORACLE_SO)).rto_cull_audit
    out = (ctypes.c_ulonglong * 5)()
    for eps, (w, h) in ((1e-6, (1, 129)), (-3e-7, (129, 1))):
        audit(ctypes.c_int(1), None)
        io, to, a, so = _render(oracle_lib, near_axis_scene(seed, 1e-6, eps), w, h, False, 2, 2, 6)
        audit(ctypes.c_int(0), out)
        for sah in (False, True):
            flagged = int(out[3]) + int(out[4]) if sah else int(out[1]) + int(out[2])   # own-box counts cover the SAH boxes
            ig, tg, b, sg = _render(hip_lib, near_axis_scene(seed, 1e-6, eps), w, h, False, 2, 2, 6, sah=sah)
            assert np.array_equal(io, ig) and np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"exact mode, sah={sah}"
            ic, tc, c, _ = _render(hip_lib, near_axis_scene(seed, 1e-6, eps), w, h, True, 2, 2, 6, sah=sah)
            differing = int((a.view(np.uint32) != c.view(np.uint32)).any(axis=2).sum())
            assert differing <= flagged, f"seed {seed} eps {eps} sah={sah}: {differing} pixels differ, the audit flags {flagged} queries"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(6)))
def test_gpu_cull_audit_equals_the_oracles_and_band_only_mode(oracle_lib, hip_lib, seed):
    """The culling audit that bench.py reports with every line (rt_stats_t.cull_audit, counted by the exact mode's
    counting launch on the GPU) is the oracle's rto_cull_audit: same queries, same own-box flags for both rules, on the
    adversarial near-axis scenes where flagged queries do exist.  And rt_set_cull(ctx, RT_CULL_BAND) - rule (a) alone -
    may differ from the oracle in at most the queries rule (a) flags (none here: it stays bit-equal)."""
    import ctypes

    from conftest import ORACLE_SO
    from raytracer_amd import ffi
    from test_oracle_cull_audit import near_axis_scene

    audit = ctypes.CDLL(str(ORACLE_SO)).rto_cull_audit
    out = (ctypes.c_ulonglong * 5)()
    total_flagged = 0
    for eps, (w, h) in ((1e-6, (1, 129)), (-3e-7, (129, 1))):
        sc = near_axis_scene(seed, 1e-6, eps)
        co = oracle_lib.create()
        co.load_scene(sc)
        co.load_camera(sc.cam)
        co.resize(w, h)
        audit(ctypes.c_int(1), None)
        co.render(1, 2, 1, 2, 6, 0)
        audit(ctypes.c_int(0), out)
        ref = co.read_mean()
        co.close()
        for sah in (False, True):
            cg = hip_lib.create()
            cg.set_accel(sah)
            cg.load_scene(sc)
            cg.load_camera(sc.cam)
            cg.resize(w, h)
            cg.set_cull(ffi.RT_CULL_OFF)
            cg.enable_stats(True)
            cg.render(1, 2, 1, 2, 6, 0)
            st = cg.stats()
            assert np.array_equal(ref.view(np.uint32), cg.read_mean().view(np.uint32))
            assert st["cull_audit"] == [int(out[0]), int(out[3]), int(out[4])], f"seed {seed} eps {eps} sah={sah}: {st['cull_audit']} vs {list(out)}"
            # rule (a) alone: differs in at most the queries its own audit flags
            cg.enable_stats(False)
            cg.set_cull(ffi.RT_CULL_BAND)
            cg.reset_accum()
            cg.render(1, 2, 1, 2, 6, 0)
            band = cg.read_mean()
            differing = int((ref.view(np.uint32) != band.view(np.uint32)).any(axis=2).sum())
            assert differing <= int(out[3]), f"seed {seed} eps {eps} sah={sah}: band-only differs in {differing} pixels, rule (a) flags {int(out[3])}"
            # a culled counting launch reports no audit (its winner is not the reference's by construction)
            cg.set_cull(ffi.RT_CULL_ON)
            cg.enable_stats(True)
            cg.reset_stats()
            cg.reset_accum()
            cg.render(1, 2, 1, 2, 6, 0)
            assert cg.stats()["cull_audit"] == [0, 0, 0]
            cg.close()
        total_flagged += int(out[3]) + int(out[4])
    print(f"seed {seed}: own-box audit flags {total_flagged} queries")


@pytest.mark.gpu
def test_set_cull_rejects_unknown_modes(hip_lib):
    from raytracer_amd.ffi import RtError

    c = hip_lib.create()
    with pytest.raises(RtError) as e:
        c.set_cull(3)
    assert e.value.code == -1
    c.close()


def _pruned_buffers(sc, seed, empty_mesh=None):
    """The scene's upload buffers with nodes the reference's builder never makes but its shader handles (:562-584): interior
    nodes that lost one child (childId = -1) and leaves emptied to zero triangles - what becomes RT_NONE in the device records."""
    from raytracer_amd.scripts import functions as Fn

    Fn.pose_dirty_meshes_on_host(sc)
    b = Fn.export_scene_buffers(sc, rest_pose=False)
    nodes = b["nodes48"].copy()
    r = np.random.default_rng(7000 + seed)
    interior = np.flatnonzero((nodes["childId1"] >= 0) & (nodes["childId2"] >= 0))
    for i in interior[r.random(len(interior)) < 0.3]:
        nodes[i]["childId2" if r.random() < 0.5 else "childId1"] = -1
    leaves = np.flatnonzero((nodes["childId1"] < 0) & (nodes["childId2"] < 0))
    for i in leaves[r.random(len(leaves)) < 0.1]:
        nodes[i]["childObjCount"] = 0
    if empty_mesh is not None:
        # a whole mesh emptied: its root becomes a leaf without triangles (an all-NaN root box on the device)
        root = int(b["meshes128"]["node0Id"][empty_mesh % len(b["meshes128"])])
        nodes[root]["childId1"] = nodes[root]["childId2"] = -1
        nodes[root]["childObjCount"] = nodes[root]["childObjOffset"] = 0
    b["nodes48"] = nodes
    return b, int(len(interior)), int((nodes["childObjCount"][leaves] == 0).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("grid", 0, (1, 129)), ("grid", 1, (129, 1)), ("grid", 2, (48, 27)), ("random", 3, (64, 36)),
                                  ("random", 5, (64, 36)), ("random", 11, (64, 36)), ("random", 3, (64, 36), 0), ("random", 5, (64, 36), 1),
                                  ("random", 11, (64, 36), 2), ("grid", 2, (48, 27), 0)])
def test_one_child_nodes_and_empty_leaves(oracle_lib, hip_lib, case):
    """Absent children carry NaN boxes on the device and the min/max slab test alone keeps rays out of them (k_pack_nodes,
    rt_dev_trace.h); rays with a zero direction component take the compare ladder, which tests the reference explicitly.  Both
    against the oracle, in every mode, on trees with one-child nodes and empty leaves, with and without a top-level tree."""
    kind, seed, (w, h) = case[:3]
    sc = _grid_scene(seed) if kind == "grid" else random_scene(seed)
    b, n_int, n_empty = _pruned_buffers(sc, seed, case[3] if len(case) > 3 else None)
    assert n_int > 3

    def run(lib, sah, cull):
        c = lib.create()
        c.set_accel(sah)
        c.set_scene(b["tris28"], b["meshes128"], b["nodes48"], b["leaf_tri_ids"], b["spheres96"], b["atmosphere80"])
        c.load_camera(sc.cam)
        c.resize(w, h)
        c.set_cull(cull)
        ids, ts = c.first_hit(1)
        c.render(1, 3, 1, 2, 5, 0)
        out = ids, ts, c.read_mean()
        c.close()
        return out

    io, to, a = run(oracle_lib, False, 0)
    assert (io >= 0).any()
    for sah in (False, True):
        for cull in (0, 1, 2):
            ig, tg, g = run(hip_lib, sah, cull)
            what = f"{kind} {seed} {w}x{h} sah={sah} cull={cull} ({n_empty} empty leaves)"
            assert np.array_equal(io, ig), f"{what}: {(io != ig).sum()} first-hit ids differ"
            assert np.array_equal(to.view(np.uint32), tg.view(np.uint32)), what
            assert np.array_equal(a.view(np.uint32), g.view(np.uint32)), f"{what}: {(a != g).any(axis=2).sum()} pixels differ"
