"""Scene model with the reference's names and byte layouts (``scripts/classes.py``), numpy only.

``Material`` / ``Sphere`` / ``Mesh`` keep their std140 ``tobytes()`` images bit-for-bit
(80 / 96 / 128 bytes) because those images are what crosses the C-ABI
(``rt_set_scene`` / ``rt_set_materials`` in include/rt_hip.h).  Packing is done with numpy
structured dtypes instead of per-object ``struct.pack`` so that a whole mesh table is one
vectorised copy.  The legacy per-object ``Triangle`` class of the reference
(classes.py:116-251, unused by its numba path) is not recreated; triangles are the SoA
``TriangleSet`` of ``numba_utils.classes``.
"""
from __future__ import annotations

from typing import Iterable

import numpy as np

from ..vecmath import Vector3, Vector4
from .animate import Animation  # noqa: F401  (re-exported like the reference)
from .numba_utils.classes import Csys, Triangle, TriangleSet  # noqa: F401

# --- std140 images ----------------------------------------------------------------------
MATERIAL_DTYPE = np.dtype(
    [
        ("color", "<f4", 4),            # @0
        ("specularColor", "<f4", 3),    # @16
        ("_pad0", "<f4"),               # @28
        ("emissionColor", "<f4", 3),    # @32
        ("emissionStrength", "<f4"),    # @44
        ("specularStrength", "<f4"),    # @48
        ("smoothness", "<f4"),          # @52
        ("transmission", "<f4"),        # @56
        ("ior", "<f4"),                 # @60
        ("metallic", "<f4"),            # @64
        ("transparentFromBehind", "<i4"),  # @68
        ("_pad1", "V8"),                # @72..79
    ]
)
assert MATERIAL_DTYPE.itemsize == 80

SPHERE_DTYPE = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("material", MATERIAL_DTYPE)])
assert SPHERE_DTYPE.itemsize == 96

MESH_DTYPE = np.dtype(
    [
        ("index", "<i4"),       # @0
        ("node0Id", "<i4"),     # @4
        ("_pad0", "V8"),
        ("bboxMin", "<f4", 4),  # @16 (w = 0)
        ("bboxMax", "<f4", 4),  # @32 (w = 0)
        ("material", MATERIAL_DTYPE),  # @48
    ]
)
assert MESH_DTYPE.itemsize == 128


class HitInfo:
    def __init__(self) -> None:
        self.didHit = False
        self.dst = np.inf
        self.hitPoint = Vector3()
        self.normal = Vector3()


class Ray:
    def __init__(self) -> None:
        self.origin = Vector3((0.0, 0.0, 0.0))
        self.dir = Vector3((0.0, 0.0, 0.0))


class Material:
    """Surface description; defaults of the reference (classes.py:36-64): magenta, ior 1.5."""

    def __init__(
        self,
        color=None,
        emissionColor=None,
        emissionStrength: float = 0.0,
        smoothness: float = 0.0,
        transmission: float = 0.0,
        ior: float = 1.5,
        transparent_from_behind=False,
        specularColor=None,
        specularStrength: float = 0.0,
        metallic: float = 0.0,
    ) -> None:
        self.color = Vector4((1.0, 0.0, 1.0, 1.0)) if color is None else color
        self.emissionColor = Vector3((0.0, 0.0, 0.0)) if emissionColor is None else emissionColor
        self.emissionStrength = emissionStrength
        self.smoothness = smoothness
        self.transmission = transmission
        self.ior = ior
        self.transparent_from_behind = transparent_from_behind
        self.metallic = metallic
        self.specularColor = Vector3((1.0, 1.0, 1.0)) if specularColor is None else specularColor
        self.specularStrength = specularStrength

    def as_record(self) -> np.ndarray:
        r = np.zeros((), dtype=MATERIAL_DTYPE)
        r["color"] = np.asarray(self.color, dtype=np.float64)
        r["specularColor"] = np.asarray(self.specularColor, dtype=np.float64)
        r["emissionColor"] = np.asarray(self.emissionColor, dtype=np.float64)
        r["emissionStrength"] = self.emissionStrength
        r["specularStrength"] = self.specularStrength
        r["smoothness"] = self.smoothness
        r["transmission"] = self.transmission
        r["ior"] = self.ior
        r["metallic"] = self.metallic
        r["transparentFromBehind"] = int(bool(self.transparent_from_behind))
        return r

    def tobytes(self) -> bytes:
        return self.as_record().tobytes()

    @property
    def opacity(self):
        return self.color.w


class Sphere:
    def __init__(self, pos, radius: float, material: Material) -> None:
        self.pos = pos
        self.radius = radius
        self.material = material

    def as_record(self) -> np.ndarray:
        r = np.zeros((), dtype=SPHERE_DTYPE)
        r["position"] = np.asarray(self.pos, dtype=np.float64)
        r["radius"] = self.radius
        r["material"] = self.material.as_record()
        return r

    def tobytes(self) -> bytes:
        return self.as_record().tobytes()


class Mesh:
    """A triangle set + pose + material + BVH graph; global registry ``Mesh.ALL`` gives ids."""

    ALL: list["Mesh"] = []

    def __init__(self, material: Material) -> None:
        self.pos = Vector3((0.0, 0.0, 0.0))
        self.triangles: TriangleSet | list = []
        self.csys = Csys()
        self.id = len(Mesh.ALL)
        Mesh.ALL.append(self)
        self.material = material
        self.bvh_graph = None
        self.is_awaiting_mesh_update = None
        self.flag_for_mesh_update()

    @classmethod
    def reset(cls) -> None:
        cls.ALL = []

    def flag_for_mesh_update(self) -> None:
        self.is_awaiting_mesh_update = True

    def unflag_for_mesh_update(self) -> None:
        self.is_awaiting_mesh_update = False

    @property
    def bounding_box(self):
        """[min, max] over the current (posed) corner positions (classes.py:386-393)."""
        pts = self.triangles.pos.reshape(-1, 3)
        return [Vector3(pts.min(axis=0), dtype=np.float32), Vector3(pts.max(axis=0), dtype=np.float32)]

    def as_record(self) -> np.ndarray:
        r = np.zeros((), dtype=MESH_DTYPE)
        lo, hi = self.bounding_box
        r["index"] = self.id
        r["node0Id"] = self.bvh_graph.first_node_id
        r["bboxMin"][:3] = lo
        r["bboxMax"][:3] = hi
        r["material"] = self.material.as_record()
        return r

    def tobytes(self) -> bytes:
        return self.as_record().tobytes()

    @classmethod
    def from_obj(cls, file: str, material: Material | None = None) -> "Mesh":
        """OBJ -> Mesh (classes.py:349-384); ids are assigned by the scene utilities."""
        from .scenes.utils import get_triangles_from_obj

        m = Mesh(material=material if material is not None else Material())
        m.triangles = get_triangles_from_obj(file, mesh_idx=m.id)
        return m


class Camera:
    """Pinhole camera parameters (classes.py:396-453); aspect fixed at 16/9."""

    def __init__(self) -> None:
        self.fov = 30.0  # degrees
        self.aspect = 16.0 / 9.0
        self.near_plane = 1000
        self.csys = Csys()
        self.antialias_strength = 0.0
        self.depth_of_field_strength = 0.0
        self.rays_per_pixel = 1
        self.bounces_per_ray = 1
        self.passes_per_frame = 1
        self.chunksx = 2
        self.chunksy = 2
        self.cycle_counter = 0

    @property
    def orientation(self):
        raise NotImplementedError

    @property
    def plane_height(self):
        return self.near_plane * np.tan(np.radians(self.fov) * 0.5) * 2.0

    @property
    def plane_width(self):
        return self.plane_height * self.aspect

    @property
    def local_to_world_matrix(self) -> np.ndarray:
        """inverse of the float64 row-vector pose matrix (classes.py:443-445)."""
        return np.linalg.inv(self.csys.transformation_matrix)

    @property
    def view_params(self) -> Vector3:
        return Vector3([self.plane_width, self.plane_height, self.near_plane])


class Scene:
    def __init__(self) -> None:
        self.meshes: list[Mesh] = []
        self.spheres: list[Sphere] = []
        self.cam = Camera()
        self.selected_objects: list = []
        self.animations: list = []
        self.atmosphere_material = Material(
            Vector4((1.0, 1.0, 1.0, 1.0), dtype="f4"), transmission=1.0, ior=1.0
        )
        self.frame_number = 0

    def validate_mesh_indices(self) -> None:
        for m in self.meshes:
            m.id = self.get_mesh_index(m)

    update_mesh_bvh_graph_map = validate_mesh_indices

    def get_mesh_index(self, msh: Mesh) -> int:
        for i, m in enumerate(self.meshes):
            if m is msh:
                return i
        raise Exception("Mesh not in list")

    def count_triangles(self) -> int:
        return sum(len(m.triangles) for m in self.meshes)

    def select_object(self, obj) -> None:
        self.selected_objects.append(obj)

    def deselect_object(self, obj) -> None:
        self.selected_objects.remove(obj)

    def animate(self, time: float) -> None:
        for a in self.animations:
            a.evaluate(time)

    def animate_frame(self, frame: int) -> None:
        for a in self.animations:
            a.evaluate(frame)


def pack_records(objs: Iterable, dtype: np.dtype) -> np.ndarray:
    """Vectorised ``iter_to_bytes``: one structured array for a list of packable objects."""
    objs = list(objs)
    out = np.zeros(len(objs), dtype=dtype)
    for i, o in enumerate(objs):
        out[i] = o.as_record()
    return out
