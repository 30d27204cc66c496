"""The animated 192-mesh dragon scene of the reference
(``scripts/scenes/dragon_scene_animated.py``): the dragon scene with the 188 cubes falling into
place over 120 frames (slow motion towards the end), camera from
``objects/monkey_blend/dragon/camera_keyframes.csv`` (120 rows), fov 25, 3 bounces, 2 rays per
pixel, 20 passes per frame (:229-259).

Same module contract as ``dragon_scene``: ``build()``, a lazily built ``scene`` and
``get_frame_number``.  What the reference's module does, restated (not what its names suggest):

* every mesh - dragon, light, floor, base and all cubes - shares ONE pose object (``csys0``,
  :62, :70-96, :113), set to ``ryg(-90)`` (:99-100);
* ``animate_cubes`` (:199-209) indexes its per-cube tables with the FRAME number, and every cube
  writes the same shared pose, so at frame i the shared pose sits at
  ``speed * (0,-1,0) * (clock(i) - clock(119))`` with speed 10 - and only the cube meshes are
  flagged for a mesh update, so only they move (the other meshes keep the geometry they were
  posed with when the scene was uploaded);
* ``clock`` is the cumulative trapezoid of a per-frame time step that eases from 1 to 0.001 of
  1/60 s over frames 30..60 and to 0 over frames 80..120 (:171-194; ``scipy.integrate.
  cumulative_trapezoid`` there, the same sum written out here);
* the per-cube random axes and angular speeds (:157-163, unseeded ``np.random``) are never used.

Fixed here as in ``dragon_scene``: cube files in sorted order, colours by ``random.Random(0)``.
"""
from __future__ import annotations

from functools import partial
from math import degrees, pi

import numpy as np

from ...vecmath import Vector3
from ..animate import FrameAnimation
from ..classes import Camera, Material, Mesh, Scene
from ..numba_utils.classes import Csys
from . import chunker
from .dragon_scene import DRAGON_DIR, _black, _rgba, cube_files, pastel_colours
from .utils import blend, get_triangles_from_obj, reset_global_datas, resolve_asset, smoothstep

KEYFRAMES_FILE = f"{DRAGON_DIR}/camera_keyframes.csv"
N_FRAMES = 120
FRAMERATE = 60
CUBE_SPEED = 10.0
CUBE_AXIS = np.array((0, -1, 0), dtype=np.float32)


def get_wall_clock_dt(i):
    """Time step of frame i (:174-185)."""
    dt = 1 / FRAMERATE
    if i < 60:
        return blend(smoothstep(30, 60, i), 1, 0.001) * dt
    return blend(smoothstep(80, 120, i), 0.001, 0.0) * dt


def wall_clock_table() -> np.ndarray:
    """clock(i) for i in 0..119: cumulative trapezoid of the time steps, initial 0 (:187-191)."""
    t = np.arange(N_FRAMES)
    v = np.array([get_wall_clock_dt(x) for x in t])
    d = np.diff(t)
    return np.concatenate([[0.0], np.cumsum(d * (v[1:] + v[:-1]) / 2.0)])


_CLOCK = wall_clock_table()


def get_wall_clock_time(i):
    return _CLOCK[i]


def cube_offset(i) -> np.ndarray:
    """Position the shared pose is given at frame i (:201-206)."""
    return np.asarray(CUBE_SPEED * CUBE_AXIS * (get_wall_clock_time(i) - get_wall_clock_time(N_FRAMES - 1))).astype(np.float32)


def animate_cubes(objs: list[Mesh], i, obj_pos_0: list[np.ndarray]):
    """:199-209 - tables indexed by the frame number i, as in the reference."""
    for obj in objs:
        obj.csys.pos = obj_pos_0[i] + cube_offset(i)
        obj.flag_for_mesh_update()


def get_frame_number(obj: Scene, i):
    """Frames 0..119, then back to 0 (:212-217)."""
    if i >= N_FRAMES:
        return 0
    obj.frame_number = i
    return obj.frame_number


def animate_camera(cam: Camera, i, keyframes: np.ndarray):
    """CSV row i -> pose and the per-frame render settings (:229-259)."""
    _, lx, ly, lz, rx, ry, rz = keyframes[i]
    cam.csys.set_pos([ly, lz, -lx])
    cam.csys.quat = np.array([0, 0, 0, 1], dtype=np.float32)
    cam.csys.rxp(degrees(ry))
    cam.csys.ryp(degrees(rz - pi / 2))
    cam.csys.rxp(degrees(rx - pi / 2))
    cam.fov = 25
    cam.depth_of_field_strength = 0.002
    cam.antialias_strength = 0.002
    cam.near_plane = 9.4
    cam.bounces_per_ray = 3
    cam.rays_per_pixel = 2
    cam.passes_per_frame = 20
    cam.chunksx = 2
    cam.chunksy = 2


def build(max_cubes: int | None = None) -> Scene:
    """Build the scene; ``max_cubes`` keeps only the first cubes (tests use a reduced scene;
    it must stay >= 120 + 0 for the frame-indexed tables, or the tables are padded)."""
    reset_global_datas()
    light = Material(_rgba(0.0, 0.0, 0.0), emissionColor=Vector3((1, 1.0, 1), dtype="f4"), emissionStrength=5.0)
    floor = Material(_rgba(0.8, 0.8, 0.8), _black(), smoothness=0.0)
    dragon = Material(_rgba(0.722, 0.457, 1.0), _black(), smoothness=0.5, specularStrength=0.2, ior=1.0)
    pose = Csys()                                   # csys0: shared by every mesh

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
    cube_meshes: list[Mesh] = []
    for name, mat in parts:
        mesh = Mesh(material=mat)
        mesh.triangles = get_triangles_from_obj(f=f"{DRAGON_DIR}/{name}", mesh_idx=mesh.id)
        mesh.csys = pose
        if name.lower().startswith("cube."):
            cube_meshes.append(mesh)
        sc.meshes.append(mesh)
        chunker.chunk_mesh_bvh(mesh)
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    sc.mesh_files = [name for name, _ in parts]
    sc.cube_meshes = cube_meshes

    keyframes = np.loadtxt(resolve_asset(KEYFRAMES_FILE), delimiter=",", skiprows=1, dtype=np.float64)
    sc.keyframes = keyframes
    # the reference's table has one entry per cube and is read at the frame number (:270, :206):
    # 188 entries cover the 120 frames; a reduced scene pads it with the same rest position
    pos0 = [m.csys.pos for m in cube_meshes]
    pos0 += [pose.pos] * max(0, N_FRAMES - len(pos0))
    sc.animations += [
        FrameAnimation(sc, get_frame_number),
        FrameAnimation(sc.cam, partial(animate_camera, keyframes=keyframes)),
        FrameAnimation(cube_meshes, partial(animate_cubes, obj_pos_0=pos0)),
    ]
    animate_camera(sc.cam, 0, keyframes)
    return sc


_scene = None


def __getattr__(name):
    global _scene
    if name == "scene":
        if _scene is None:
            _scene = build()
        return _scene
    raise AttributeError(name)
