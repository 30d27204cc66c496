"""Buffer helpers of the reference's ``scripts/functions.py`` that the render path uses.

Only ``iter_to_bytes`` (functions.py:54-59) and the screen->ray helper
(``convert_screen_coords_to_camera_ray``, :251-278) are on or beside the path; the
reference's single-ray picking code is dead (its ``rayTriangle`` returns an empty hit on its
first statement, :88) and is not recreated.  ``export_scene_buffers`` /
``camera_uniforms`` gather exactly the byte images and uniform values that
``RayTracerDynamic.configure_buffers_and_uniforms`` / ``calculate_frame`` upload
(application.py:919-1002, 1175-1183) so they can cross the C-ABI in one call.
"""
from __future__ import annotations

from typing import Iterable

import numpy as np

from .classes import MATERIAL_DTYPE, MESH_DTYPE, SPHERE_DTYPE, Camera, Mesh, Scene, pack_records
from .numba_utils import classes as _nb
from .scenes import chunker


def iter_to_bytes(iterable: Iterable) -> bytearray:
    out = bytearray()
    for x in iterable:
        out.extend(x.tobytes())
    return out


def pose_dirty_meshes_on_host(scene: Scene) -> list[int]:
    """Host twin of application.py:939-952: pose flagged meshes, refit their graphs.
    Returns the ids of the meshes that were dirty."""
    dirty = []
    for mesh in scene.meshes:
        if mesh.is_awaiting_mesh_update:
            _nb.update_triangles_to_csys(mesh.triangles, mesh.csys)
            mesh.bvh_graph.flag_for_mesh_update()
            mesh.unflag_for_mesh_update()
            dirty.append(mesh.id)
    for graph in chunker.BVHGraph.ALL:
        graph.update_graph_node_aabbs_for_changed_triangles(force=False)
    return dirty


def export_scene_buffers(scene: Scene, rest_pose: bool = False) -> dict:
    """The six buffers of SURVEY.md 3.4 as numpy arrays (std140 images).

    rest_pose=True packs the un-posed triangles (the device then applies poses itself via
    ``rt_update_mesh_pose``); False packs current host positions like the reference does.
    """
    reg = _nb.get_all_triangles_arr()
    if rest_pose:
        saved = [(s.pos, s.nrm) for s in reg.sets]
        for s in reg.sets:
            s.pos, s.nrm = s.pos0, s.nrm0
        tris = _nb.triangles_to_array(reg)
        for s, (p, n) in zip(reg.sets, saved):
            s.pos, s.nrm = p, n
    else:
        tris = _nb.triangles_to_array(reg)
    return {
        "tris28": tris,
        "meshes128": pack_records(Mesh.ALL, MESH_DTYPE),
        "nodes48": chunker.BVHGraph.all_node_records(),
        "leaf_tri_ids": np.asarray(chunker.BVHGraph.BVH_TRI_ID_LIST_GLOBAL, dtype=np.int32),
        "spheres96": pack_records(scene.spheres, SPHERE_DTYPE),
        "atmosphere80": scene.atmosphere_material.as_record().reshape(1),
    }


def camera_uniforms(cam: Camera) -> dict:
    """Values written per frame at application.py:1175-1183, cast to float32 the same way."""
    return {
        "view_params": np.asarray(cam.view_params, dtype=np.float64).astype(np.float32),
        "local_to_world16": np.ascontiguousarray(cam.local_to_world_matrix.astype(np.float32)).reshape(16),
        "cam_pos": np.asarray(cam.csys.pos).astype(np.float32),
        "dof": np.float32(cam.depth_of_field_strength),
        "aa": np.float32(cam.antialias_strength),
    }


def convert_screen_coords_to_camera_ray(x: float, y: float, width: int, height: int, cam: Camera):
    """Pixel -> (origin, unit direction) with the shader's camera maths (raytracer.fs.glsl:961-967)."""
    u = camera_uniforms(cam)
    ndc = np.array([(2.0 * (x + 0.5)) / width - 1.0, (2.0 * (y + 0.5)) / height - 1.0])
    local = np.array([ndc[0] / 2.0, ndc[1] / 2.0, 1.0]) * u["view_params"].astype(np.float64)
    m = u["local_to_world16"].reshape(4, 4).astype(np.float64)
    world = m[:3, :3] @ local + m[:3, 3]
    return u["cam_pos"].astype(np.float64), world / np.linalg.norm(world)
