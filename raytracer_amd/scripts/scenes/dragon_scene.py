"""The 192-mesh dragon scene of the reference (``scripts/scenes/dragon_scene.py``) -
BASELINE.json configs[4]: glossy dragon (10 000 triangles), base, floor, emissive light and
188 pastel cubes of 12 triangles, 8 bounces.

Same module contract as ``final_scene_numba``: ``build()``, ``scene`` (built lazily on first
attribute access, because the BVH of the dragon takes a few seconds in numpy) and
``get_frame_number``.

Fixed here because the reference leaves it to the file system: the cube
files are taken in *sorted* order (the reference iterates ``Path.glob`` order,
dragon_scene.py:87), and the pastel colour of cube k is the k-th ``random.choice`` after
``random.seed(0)`` in that order (:76-94).  Every mesh shares one pose, ``ryg(-90)`` (:97-98).
"""
from __future__ import annotations

import random
from math import degrees, pi

import numpy as np

from ...vecmath import Vector3, Vector4
from ..animate import FrameAnimation
from ..classes import Camera, Material, Mesh, Scene
from ..numba_utils.classes import Csys
from . import chunker
from .utils import get_triangles_from_obj, reset_global_datas, resolve_asset

DRAGON_DIR = "objects/monkey_blend/dragon"
# camera of the reference's hook (:123-132): Blender location / XYZ euler of the single view
CAM_LOC = (9.432361602783203, 0.059754848, 0.9540680646896362)
CAM_ROT = (1.5707963705062866, 0.0, 1.5707963705062866)


def _rgba(r, g, b, a=1.0):
    return Vector4((r, g, b, a), dtype="f4")


def _black():
    return Vector3((0.0, 0.0, 0.0), dtype="f4")


def cube_files() -> list[str]:
    d = resolve_asset(DRAGON_DIR)
    return sorted(p.name for p in d.glob("Cube.*.obj"))


def pastel_colours(n_cubes: int) -> list[tuple[float, float, float]]:
    """random.seed(0); six pastels; one random.choice per cube (:76-94)."""
    rng = random.Random(0)
    palette = [(rng.uniform(0.7, 1.0), rng.uniform(0.7, 1.0), rng.uniform(0.7, 1.0)) for _ in range(6)]
    return [rng.choice(palette) for _ in range(n_cubes)]


def animate_camera_params(cam: Camera, i: int):
    """One fixed view; fov 20, AA 0.002, 8 bounces (:123-145)."""
    cam.csys.set_pos([CAM_LOC[1], CAM_LOC[2], -CAM_LOC[0]])
    cam.csys.quat = np.array([0, 0, 0, 1], dtype=np.float32)
    cam.csys.rxp(degrees(CAM_ROT[1]))
    cam.csys.ryp(degrees(CAM_ROT[2] - pi / 2))
    cam.csys.rxp(degrees(CAM_ROT[0] - pi / 2))
    cam.fov = 20
    cam.depth_of_field_strength = 0.00
    cam.antialias_strength = 0.002
    cam.bounces_per_ray = 8
    cam.rays_per_pixel = 1
    cam.passes_per_frame = 10000
    cam.chunksx = 2
    cam.chunksy = 2


def get_frame_number(obj: Scene, i):
    """The reference pins the frame to 0 (:149-152): a still."""
    obj.frame_number = 0
    return obj.frame_number


def build(max_cubes: int | None = None) -> Scene:
    """Build the scene; ``max_cubes`` keeps only the first cubes (tests use a reduced scene)."""
    reset_global_datas()
    light = Material(_rgba(0.0, 0.0, 0.0), emissionColor=Vector3((1, 1.0, 1), dtype="f4"), emissionStrength=5.0)
    floor = Material(_rgba(0.8, 0.8, 0.8), _black(), smoothness=0.0)
    dragon = Material(_rgba(0.722, 0.457, 1.0), _black(), smoothness=0.5, specularStrength=0.2, ior=1.0)
    pose = Csys()

    parts = [("dragon-3.obj", dragon), ("light.obj", light), ("floor.obj", floor), ("base.obj", floor)]
    cubes = cube_files()
    colours = pastel_colours(len(cubes))
    if max_cubes is not None:
        cubes, colours = cubes[:max_cubes], colours[:max_cubes]
    parts += [(name, Material(_rgba(*rgb), _black(), smoothness=0.0)) for name, rgb in zip(cubes, colours)]

    pose._pos = np.array([0.0, 0.0, 0.0], dtype=np.float32)
    pose.ryg(-90)

    sc = Scene()
    sc.atmosphere_material = Material(_rgba(0.5, 0.5, 0.5), transmission=1.0, ior=1.0)
    for name, mat in parts:
        mesh = Mesh(material=mat)
        mesh.triangles = get_triangles_from_obj(f=f"{DRAGON_DIR}/{name}", mesh_idx=mesh.id)
        mesh.csys = pose
        sc.meshes.append(mesh)
        chunker.chunk_mesh_bvh(mesh)
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    sc.mesh_files = [name for name, _ in parts]
    sc.animations += [FrameAnimation(sc, get_frame_number), FrameAnimation(sc.cam, animate_camera_params)]
    animate_camera_params(sc.cam, 0)
    return sc


_scene = None


def __getattr__(name):
    global _scene
    if name == "scene":
        if _scene is None:
            _scene = build()
        return _scene
    raise AttributeError(name)
