"""Host-side mirror of the reference's ``scripts/numba_utils/classes.py`` (numpy, no numba).

Same public names and argument meaning as the reference (``Csys``, ``Triangle``,
``update_triangles_to_csys``, ``triangles_to_array``,
``create_and_register_triangles_from_obj_data``, the ``ALL_TRIANGLES`` registry helpers),
but the storage is MI355X-first: a mesh's triangles live in ONE structure-of-arrays block
(``TriangleSet``: rest and posed positions/normals as (n,3,3) float32) that can be handed
to the device in a single copy, instead of a Python list of per-triangle heap objects.
``Triangle`` is a thin indexed view onto such a block so ``for t in mesh.triangles`` code
keeps working.

The pose transform itself is a device kernel in the product path
(``rt_update_mesh_pose``, include/rt_hip.h); ``update_triangles_to_csys`` below is the
host twin used for BVH construction, picking helpers and CPU-side checks.  Its arithmetic
is the build's fixed definition of the reference's ``np.dot([p,1], T32)``
(numba_utils/classes.py:195-224, BLAS order unspecified there): a k-ordered float32 sum
``((p.x*T0j + p.y*T1j) + p.z*T2j) + T3j`` with no fused multiply-add.
"""
from __future__ import annotations

from typing import Iterable, Iterator, Sequence

import numpy as np

from .functions import (
    mat33_create_from_quaternion,
    mat44_create_from_quaternion,
    multiply_quaternions,
    normalize,
    quaternion_create_from_axis_rotation,
)

_F = np.float32


class Csys:
    """Pose = float32 position + float32 quaternion (x, y, z, w); classes.py:24-119."""

    __slots__ = ("_pos", "quat")

    def __init__(self) -> None:
        self._pos = np.zeros(3, dtype=_F)
        self.quat = np.array([0, 0, 0, 1], dtype=_F)

    # -- accessors ---------------------------------------------------------------------
    @property
    def pos(self) -> np.ndarray:
        return self._pos

    @pos.setter
    def pos(self, val) -> None:
        self._pos = np.asarray(val)

    @property
    def rotation_matrix(self) -> np.ndarray:
        return mat33_create_from_quaternion(self.quat)

    @property
    def transformation_matrix(self) -> np.ndarray:
        """float64 4x4, rotation in [0:3,0:3], position in ROW 3 (row-vector convention)."""
        m = mat44_create_from_quaternion(self.quat)
        m[3, 0:3] = self._pos
        return m

    def set_pos(self, p) -> "Csys":
        v = np.array(p, dtype=_F)
        if v.shape != (3,):
            raise ValueError("must provide (3,) vector")
        self._pos = v
        return self

    # -- rotations: quat <- normalize(quat * q_axis), right-multiplication --------------
    def _spin(self, axis: np.ndarray, degrees: float) -> "Csys":
        q = quaternion_create_from_axis_rotation(np.asarray(axis, dtype=_F), np.radians(degrees))
        self.quat = normalize(multiply_quaternions(self.quat, q))
        return self

    def rxg(self, degrees: float) -> "Csys":
        return self._spin((1.0, 0.0, 0.0), degrees)

    def ryg(self, degrees: float) -> "Csys":
        return self._spin((0.0, 1.0, 0.0), degrees)

    def rzg(self, degrees: float) -> "Csys":
        return self._spin((0.0, 0.0, 1.0), degrees)

    def rxp(self, degrees: float) -> "Csys":
        return self._spin(self.rotation_matrix[0][0:3], degrees)

    def ryp(self, degrees: float) -> "Csys":
        return self._spin(self.rotation_matrix[1][0:3], degrees)

    def rzp(self, degrees: float) -> "Csys":
        return self._spin(self.rotation_matrix[2][0:3], degrees)

    # -- translations ------------------------------------------------------------------
    def tg(self, pos: np.ndarray) -> None:
        # the reference forgets ``return self`` here (classes.py:94-95); kept
        self._pos = np.asarray(pos).astype(_F)

    def _shift_global(self, k: int, dst: float) -> "Csys":
        self._pos[k] += dst
        return self

    def txg(self, dst: float) -> "Csys":
        return self._shift_global(0, dst)

    def tyg(self, dst: float) -> "Csys":
        return self._shift_global(1, dst)

    def tzg(self, dst: float) -> "Csys":
        return self._shift_global(2, dst)

    def _shift_local(self, k: int, dst: float) -> "Csys":
        self._pos = (self._pos + dst * self.rotation_matrix[k][0:3]).astype(_F)
        return self

    def txp(self, dst: float) -> "Csys":
        return self._shift_local(0, dst)

    def typ(self, dst: float) -> "Csys":
        return self._shift_local(1, dst)

    def tzp(self, dst: float) -> "Csys":
        return self._shift_local(2, dst)


# ----------------------------------------------------------------------------------------
# triangle storage
# ----------------------------------------------------------------------------------------
class TriangleSet(Sequence):
    """SoA block of one mesh's triangles; behaves like ``list[Triangle]``.

    pos0/nrm0 : (n,3,3) float32 rest pose (corner, xyz); pos/nrm : current pose.
    ids       : (n,) int32 global triangle ids (== index into the global triangle array).
    """

    def __init__(self, pos0: np.ndarray, nrm0: np.ndarray, mesh_index: int, id_start: int):
        n = pos0.shape[0]
        assert pos0.shape == (n, 3, 3) and nrm0.shape == (n, 3, 3)
        self.pos0 = np.ascontiguousarray(pos0, dtype=_F)
        self.nrm0 = np.ascontiguousarray(nrm0, dtype=_F)
        self.pos = self.pos0.copy()
        self.nrm = self.nrm0.copy()
        self.mesh_index = int(mesh_index)
        self.ids = np.arange(id_start, id_start + n, dtype=np.int32)

    def __len__(self) -> int:
        return self.pos0.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [Triangle._view(self, j) for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return Triangle._view(self, i)

    def __iter__(self) -> Iterator["Triangle"]:
        for i in range(len(self)):
            yield Triangle._view(self, i)

    def __bool__(self) -> bool:
        return len(self) > 0


class Triangle:
    """One triangle; either a view into a TriangleSet or a standalone one (classes.py:158-232)."""

    __slots__ = ("_set", "_i")

    def __init__(self, posA, posB, posC, normalA, normalB, normalC, mesh_parent_index: int, triangle_id: int):
        p = np.array([posA, posB, posC], dtype=_F)[None]
        n = np.array([normalA, normalB, normalC], dtype=_F)[None]
        self._set = TriangleSet(p, n, mesh_parent_index, triangle_id)
        self._i = 0

    @classmethod
    def _view(cls, s: TriangleSet, i: int) -> "Triangle":
        t = object.__new__(cls)
        t._set = s
        t._i = i
        return t

    posA = property(lambda s: s._set.pos[s._i, 0])
    posB = property(lambda s: s._set.pos[s._i, 1])
    posC = property(lambda s: s._set.pos[s._i, 2])
    posA0 = property(lambda s: s._set.pos0[s._i, 0])
    posB0 = property(lambda s: s._set.pos0[s._i, 1])
    posC0 = property(lambda s: s._set.pos0[s._i, 2])
    normalA = property(lambda s: s._set.nrm[s._i, 0])
    normalB = property(lambda s: s._set.nrm[s._i, 1])
    normalC = property(lambda s: s._set.nrm[s._i, 2])
    normalA0 = property(lambda s: s._set.nrm0[s._i, 0])
    normalB0 = property(lambda s: s._set.nrm0[s._i, 1])
    normalC0 = property(lambda s: s._set.nrm0[s._i, 2])
    mesh_index = property(lambda s: s._set.mesh_index)
    id = property(lambda s: int(s._set.ids[s._i]))

    @property
    def positions(self):
        return self.posA, self.posB, self.posC

    @property
    def normals(self):
        return self.normalA, self.normalB, self.normalC

    def update_pos_to_csys(self, csys: Csys) -> "Triangle":
        _pose_rows(self._set, csys, slice(self._i, self._i + 1))
        return self


def _pose_rows(s: TriangleSet, csys: Csys, rows=slice(None)) -> None:
    t = csys.transformation_matrix.astype(_F)
    p = s.pos0[rows]
    n = s.nrm0[rows]
    px, py, pz = p[..., 0], p[..., 1], p[..., 2]
    nx, ny, nz = n[..., 0], n[..., 1], n[..., 2]
    for j in range(3):
        s.pos[rows, :, j] = ((px * t[0, j] + py * t[1, j]) + pz * t[2, j]) + t[3, j]
        s.nrm[rows, :, j] = (nx * t[0, j] + ny * t[1, j]) + nz * t[2, j]


# -- global registry (reference: module-global ALL_TRIANGLES list, classes.py:141-152) ------
class _Registry(Sequence):
    def __init__(self):
        self.sets: list[TriangleSet] = []

    def __len__(self):
        return sum(len(s) for s in self.sets)

    def __getitem__(self, i):
        if i < 0:
            i += len(self)
        for s in self.sets:
            if i < len(s):
                return s[i]
            i -= len(s)
        raise IndexError

    def __iter__(self):
        for s in self.sets:
            yield from s


ALL_TRIANGLES = _Registry()


def add_to_all_triangles(l) -> None:
    if isinstance(l, TriangleSet):
        ALL_TRIANGLES.sets.append(l)
    else:  # a plain iterable of standalone triangles
        for t in l:
            ALL_TRIANGLES.sets.append(t._set)


def reset_all_triangles() -> None:
    ALL_TRIANGLES.sets = []


def get_all_triangles_arr() -> _Registry:
    return ALL_TRIANGLES


def _sets_of(triangles) -> list[TriangleSet]:
    if isinstance(triangles, TriangleSet):
        return [triangles]
    if isinstance(triangles, _Registry):
        return triangles.sets
    out: list[TriangleSet] = []
    for t in triangles:
        if not out or out[-1] is not t._set:
            out.append(t._set)
    return out


def update_triangles_to_csys(triangles, csys: Csys) -> None:
    """Pose every triangle from its REST copy (no drift); classes.py:254-257."""
    for s in _sets_of(triangles):
        _pose_rows(s, csys)


def triangles_to_array(triangles) -> np.ndarray:
    """N x 28 float32 std140 image of ``Triangle`` (classes.py:315-342, shader :93-104).

    f32 slots: A 0-2, B 4-6, C 8-10, nA 12-14, nB 16-18, nC 20-22, [23] = mesh index bits,
    [24] = triangle id bits, the rest zero.
    """
    sets = _sets_of(triangles)
    n = sum(len(s) for s in sets)
    out = np.zeros((n, 28), dtype=_F)
    as_i32 = out.view(np.int32)
    at = 0
    for s in sets:
        m = len(s)
        blk = out[at:at + m]
        for c in range(3):
            blk[:, 4 * c:4 * c + 3] = s.pos[:, c]
            blk[:, 12 + 4 * c:15 + 4 * c] = s.nrm[:, c]
        as_i32[at:at + m, 23] = s.mesh_index
        as_i32[at:at + m, 24] = s.ids
        at += m
    return out


def create_and_register_triangles_from_obj_data(
    vertex_indices_arr: np.ndarray,
    vertices: np.ndarray,
    vertex_normals: np.ndarray,
    mesh_idx: int,
    triangle_id_start: int,
) -> TriangleSet:
    """faces (n,3) int + per-vertex positions/normals -> TriangleSet; ids start..start+n-1
    (classes.py:345-365).  Does not touch the registry (the caller does, scenes/utils.py:29)."""
    f = np.asarray(vertex_indices_arr, dtype=np.int64)
    return TriangleSet(
        np.asarray(vertices, dtype=_F)[f],
        np.asarray(vertex_normals, dtype=_F)[f],
        mesh_idx,
        triangle_id_start,
    )
