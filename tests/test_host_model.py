"""Host-side mirror of the reference's scene model: pose algebra, BVH builder, packers."""
import numpy as np
import pytest

from conftest import build_suzanne, make_ctx

F = np.float32


def test_csys_pose_algebra():
    from raytracer_amd.scripts.numba_utils.classes import Csys

    c = Csys()
    assert c.quat.dtype == F and np.array_equal(c.quat, [0, 0, 0, 1])
    c.ryg(90)
    R = c.rotation_matrix
    assert R.dtype == F and np.allclose(R @ R.T, np.eye(3), atol=1e-6)
    # row-vector convention (numba_utils/classes.py:44-48): world = p . R ; row 0 = (m00, m01, m02) with
    # m02 = 2(xz+yw) = sin(90deg) for q = (0, sin45, 0, cos45): +x goes to +z
    assert np.allclose(np.array([1, 0, 0], dtype=F) @ R, [0, 0, 1], atol=1e-6)
    T = c.set_pos([1, 2, 3]).transformation_matrix
    assert T.dtype == np.float64 and np.array_equal(T[3, :3], [1, 2, 3]) and T[3, 3] == 1
    # right-multiplication: rxp uses the CURRENT local x axis
    a = Csys().ryg(90).rxp(30)
    b = Csys().ryg(90)
    b.rxg(0)  # no-op keeps normalisation path exercised
    assert not np.allclose(a.quat, b.quat)
    assert np.isclose(np.linalg.norm(a.quat), 1.0, atol=1e-6)
    # tg forgets to return self in the reference (classes.py:94-95)
    assert Csys().tg(np.array([1.0, 2.0, 3.0])) is None


def test_obj_reader_counts():
    from raytracer_amd.scripts.scenes.utils import read_obj

    pos, nrm = read_obj("objects/monkey_blend/monkey.obj")
    assert pos.shape == (968, 3, 3) and nrm.shape == (968, 3, 3) and pos.dtype == F
    assert np.allclose(np.linalg.norm(nrm, axis=2), 1.0, atol=1e-3)
    assert read_obj("objects/monkey_blend/wall_front.obj")[0].shape[0] == 2


def test_suzanne_scene_structure(suzanne):
    from raytracer_amd.scripts import functions as Fn
    from raytracer_amd.scripts.scenes import chunker

    assert suzanne.count_triangles() == 1052 and len(suzanne.meshes) == 9
    nodes = chunker.BVHParentNode.ALL
    assert all(n.id == i for i, n in enumerate(nodes))
    for n in nodes:
        if n.is_leaf():
            assert n.tris_start_offset is not None
        else:
            assert n.child_left.id == n.child_right.id - 1 > n.id
            assert n.tris_count == n.child_left.tris_count + n.child_right.tris_count
            assert n.tris_count > 5 and n.depth < 20
    ids = np.asarray(chunker.BVHGraph.BVH_TRI_ID_LIST_GLOBAL)
    assert sorted(ids.tolist()) == list(range(1052))                               # every triangle in exactly one leaf
    b = Fn.export_scene_buffers(suzanne)
    assert b["tris28"].shape == (1052, 28) and b["meshes128"].nbytes == 9 * 128 and b["nodes48"].nbytes == len(nodes) * 48
    assert np.array_equal(b["tris28"].view(np.int32)[:, 24], np.arange(1052))
    # mesh record points at its graph's first node; node record of an interior node has offset -1
    assert [int(m["node0Id"]) for m in b["meshes128"]] == [m.bvh_graph.first_node_id for m in suzanne.meshes]
    inter = b["nodes48"][b["nodes48"]["childId1"] >= 0]
    assert (inter["childObjOffset"] == -1).all() and (inter["childObjCount"] > 5).all()


def test_cube_bvh_topology():
    """12-triangle axis-aligned slab (wall_top.obj): hand-checkable split (chunker.py:216-244)."""
    from raytracer_amd.scripts.classes import Material, Mesh
    from raytracer_amd.scripts.scenes import chunker
    from raytracer_amd.scripts.scenes.utils import get_triangles_from_obj, reset_global_datas

    reset_global_datas()
    m = Mesh(Material())
    m.triangles = get_triangles_from_obj("objects/monkey_blend/wall_top.obj", m.id)
    chunker.chunk_mesh_bvh(m)
    root = m.bvh_graph.nodes[0]
    assert root.tris_count == 12 and not root.is_leaf()
    # longest axis of the 4 x 0.2 x 4.4 slab is z: left child = centroids with z <= mid
    lo, hi = root.aabb
    assert int(np.argmax(np.abs(lo - hi))) == 2
    mid = hi[2] - (hi[2] - lo[2]) * F(0.5)
    left = [i for i in range(12) if root.centroids[i][2] <= mid]
    assert root.child_left.tri_idx.tolist() == left
    assert sorted(root.child_left.tri_idx.tolist() + root.child_right.tri_idx.tolist()) == list(range(12))
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    leaves = [n for n in m.bvh_graph.nodes if n.is_leaf()]
    # a node whose split would leave one side empty stays an oversized leaf (chunker.py:236-239)
    assert sum(n.tris_count for n in leaves) == 12 and all(n.tris_count <= 5 or n.depth > 0 for n in leaves)


def test_host_pose_and_refit_equal_oracle(oracle_lib, suzanne):
    """numpy update_triangles_to_csys / chunker refit == the oracle's rt_update_mesh_pose, bit for bit
    (both restate numba_utils/classes.py:195-224 and chunker.py:197-214)."""
    from raytracer_amd.scripts import functions as Fn
    from raytracer_amd.scripts.scenes.chunker import NODE_DTYPE

    ctx = make_ctx(oracle_lib, suzanne, 16, 9, rest_pose=True)
    dev_tris, dev_nodes = ctx.read_tris28(), ctx.read_nodes48(NODE_DTYPE)
    for m in suzanne.meshes:
        m.flag_for_mesh_update()
    Fn.pose_dirty_meshes_on_host(suzanne)
    host = Fn.export_scene_buffers(suzanne, rest_pose=False)
    assert np.array_equal(host["tris28"].view(np.uint32), dev_tris.view(np.uint32))
    assert host["nodes48"].tobytes() == dev_nodes.tobytes()


def test_material_animation_and_keyframes():
    sc = build_suzanne(0)
    front = sc.materials["front"]
    assert front.transparent_from_behind is True and float(front.color[3]) == 0.0      # alpha ramps from 0 (:257-283)
    sc.animate_frame(65)
    assert 0.0 < float(sc.materials["front"].color[3]) < 1.0
    sc.animate_frame(249)
    assert sc.frame_number == 249
    assert sc.cam.fov == 15 and sc.cam.antialias_strength == 0.001
    assert abs(float(sc.cam.near_plane) - np.linalg.norm(2 * (sc.meshes[0].csys.pos - sc.cam.csys.pos))) < 1e-4


def check_error_codes(lib, suzanne):
    """The error behaviour of the C-ABI (include/rt_hip.h): same codes from the product and the oracle."""
    from raytracer_amd.ffi import RtError
    from raytracer_amd.scripts import functions as Fn

    codes = []

    def expect(code, fn, *a):
        with pytest.raises(RtError) as e:
            fn(*a)
        assert e.value.code == code, (fn.__name__, e.value.code, str(e.value))
        codes.append(e.value.code)
        return str(e.value)

    ctx = lib.create()
    expect(-2, ctx.render, 1, 1, 1, 1, 1, 0)                       # RT_E_STATE: no scene yet
    b = Fn.export_scene_buffers(suzanne, rest_pose=True)
    args = lambda **kw: [kw.get(k, b[k]) for k in ("tris28", "meshes128", "nodes48", "leaf_tri_ids")] + [None, b["atmosphere80"]]
    bad = b["leaf_tri_ids"].copy()
    bad[3] = 999999
    assert "leaf_tri_ids[3]" in expect(-1, ctx.set_scene, *args(leaf_tri_ids=bad))
    bad = b["tris28"].copy()
    bad.view(np.int32)[5, 24] = 7                                  # triId != index
    expect(-1, ctx.set_scene, *args(tris28=bad))
    bad = b["nodes48"].copy()
    bad["childId1"][0] = 0                                         # a node that is its own child
    expect(-1, ctx.set_scene, *args(nodes48=bad))
    bad = b["meshes128"].copy()
    bad["node0Id"][2] = len(b["nodes48"]) + 5
    expect(-1, ctx.set_scene, *args(meshes128=bad))
    expect(-4, ctx.set_scene, b["tris28"], np.repeat(b["meshes128"], 40, axis=0), b["nodes48"], b["leaf_tri_ids"], None,
           b["atmosphere80"])                                      # RT_E_LIMIT: 360 meshes
    # a chain deeper than RT_MAX_STACK: node i has the single child i+1, the last one is the leaf
    from raytracer_amd.scripts.scenes.chunker import NODE_DTYPE
    deep = np.zeros(40, dtype=NODE_DTYPE)
    deep["id"] = np.arange(40)
    deep["childId1"] = np.arange(1, 41)
    deep["childId2"] = -1
    deep["childId1"][-1] = -1
    deep["childObjOffset"] = -1
    deep["childObjOffset"][-1] = 0
    deep["childObjCount"][-1] = 1
    one_mesh = b["meshes128"][:1].copy()
    one_mesh["node0Id"] = 0
    tri0 = b["tris28"][:1].copy()
    expect(-4, ctx.set_scene, tri0, one_mesh, deep, np.zeros(1, np.int32), None, b["atmosphere80"])
    # a triangle listed in two leaves (the reference's builder never does that, chunker.py:216-244)
    bad = b["leaf_tri_ids"].copy()
    bad[1] = bad[0]
    assert "more than one leaf" in expect(-1, ctx.set_scene, *args(leaf_tri_ids=bad))
    expect(-2, ctx.bind_accum, 12345)                             # no image size yet: nothing to bind to
    ctx.set_scene(*args())
    expect(-1, ctx.update_mesh_pose, 99, [0, 0, 0], [0, 0, 0, 1])
    expect(-2, ctx.render, 1, 1, 1, 1, 1, 0)                       # no camera
    ctx.load_camera(suzanne.cam)
    expect(-2, ctx.render, 1, 1, 1, 1, 1, 0)                       # no image size
    expect(-1, ctx.resize, 0, 4)
    expect(-1, ctx.resize, 40000, 4)
    ctx.resize(8, 4)
    expect(-1, ctx.render, 1, 2, 3, 1, 1, 0)                       # running mean needs stride 1
    expect(-1, ctx.render, 1, 2, 1, 0, 1, 0)                       # rays_per_pixel < 1
    expect(-1, ctx.render, 1, 2, 1, 1, -1, 0)                      # negative bounces
    expect(-1, ctx.render, 1, 2, 1, 1, 1, 7)                       # unknown flags
    expect(-1, ctx.render_static, 1, 2, 3, 1, 1, 0)                # the allocation-free variant checks the same way
    ctx.render(1, 2, 1, 1, 1, 0)                                   # and after all that the context still works
    assert np.isfinite(ctx.read_mean()).all()
    # MAX_CYCLES (raytracer.fs.glsl:931-935): passes beyond it leave the image as it is
    img2 = ctx.read_mean()
    ctx.set_max_cycles(2)
    ctx.render(3, 4, 1, 1, 1, 0)
    assert np.array_equal(ctx.read_mean().view(np.uint32), img2.view(np.uint32)) and ctx.read_accum()[1] == 2
    ctx.reset_accum()
    ctx.render(1, 5, 1, 1, 1, 0)                                   # passes 1, 2 count; 3, 4, 5 do not
    assert np.array_equal(ctx.read_mean().view(np.uint32), img2.view(np.uint32)) and ctx.read_accum()[1] == 2
    ctx.set_max_cycles(0xFFFFFFFF)
    ctx.close()
    return codes


def test_error_codes(oracle_lib, suzanne):
    assert len(check_error_codes(oracle_lib, suzanne)) == 19
