"""Minimal stand-ins for the pyrr types the reference's scene modules construct.

The reference builds materials with ``pyrr.Vector4((r, g, b, a), dtype="f4")`` and reads
``.xyz``, ``.w``, ``.normalized``, ``.squared_length`` (scripts/scenes/final_scene_numba.py:31-122,
311-330; scripts/classes.py:36-104).  pyrr is not installable here, so these ndarray
subclasses provide exactly that surface and nothing else.  Arithmetic is plain numpy.
"""
from __future__ import annotations

import numpy as np


class _Vec(np.ndarray):
    _n = 0

    def __new__(cls, value=None, dtype=None):
        if value is None:
            value = (0.0,) * cls._n
        arr = np.asarray(value, dtype=dtype if dtype is not None else np.float64)
        if arr.shape != (cls._n,):
            raise ValueError(f"{cls.__name__} needs {cls._n} components, got shape {arr.shape}")
        return arr.copy().view(cls)

    def __array_finalize__(self, obj):  # keep subclass through arithmetic of same shape only
        pass

    def __array_wrap__(self, out, context=None, return_scalar=False):
        if isinstance(out, np.ndarray) and out.shape == (self._n,):
            return out.view(type(self))
        out = np.asarray(out)
        return out[()] if out.ndim == 0 else out

    @property
    def x(self):
        return float(np.ndarray.__getitem__(self, 0))

    @property
    def y(self):
        return float(np.ndarray.__getitem__(self, 1))

    @property
    def z(self):
        return float(np.ndarray.__getitem__(self, 2))

    @property
    def squared_length(self):
        a = np.asarray(self)
        return float(np.dot(a, a))

    @property
    def length(self):
        return float(np.sqrt(self.squared_length))

    @property
    def normalized(self):
        a = np.asarray(self)
        return (a / np.sqrt(np.dot(a, a))).view(type(self))

    def normalize(self):
        a = np.asarray(self)
        a /= np.sqrt(np.dot(a, a))


class Vector3(_Vec):
    _n = 3

    def cross(self, other):
        return np.cross(np.asarray(self), np.asarray(other)).view(Vector3)


class Vector4(_Vec):
    _n = 4

    @property
    def w(self):
        return float(np.ndarray.__getitem__(self, 3))

    @property
    def xyz(self):
        return np.asarray(self)[:3].copy().view(Vector3)
