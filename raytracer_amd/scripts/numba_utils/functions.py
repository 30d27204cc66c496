"""Pose algebra of the reference's ``scripts/numba_utils/functions.py:26-128``, numpy only.

numba is not part of this build; what matters for the renderer is the *arithmetic* those
jitted helpers perform, because the camera matrix and every posed triangle flow from it.
The dtype rules below restate what numba's typing gives the reference:

* quaternions/axes are float32 arrays ordered x, y, z, w;
* products of two float32 scalars stay float32; ``1 / f32`` and ``2.0 * f32`` promote to
  float64; every returned array is cast back to the input dtype.

parity unpinned: nothing in the reference pins these roundings (no tests, numba absent).
"""
from __future__ import annotations

import time
from functools import wraps
from typing import Callable

import numpy as np

_F = np.float32


def timer(fn: Callable):
    """Wall-clock decorator with the reference's print format (functions.py:14-23)."""

    @wraps(fn)
    def timed(*a, **kw):
        t0 = time.time()
        out = fn(*a, **kw)
        print(f"Function {fn.__name__} executed in {(time.time() - t0) * 1e3:.2f} ms")
        return out

    return timed


def _norm(v: np.ndarray):
    # sqrt of the k-ordered sum of squares, kept in the vector's dtype
    v = np.asarray(v)
    acc = v.dtype.type(0)
    for c in v:
        acc = acc + c * c
    return np.sqrt(acc)


def normalize(vec: np.ndarray) -> np.ndarray:
    """vec / |vec| in vec's dtype (functions.py:26-28)."""
    vec = np.asarray(vec)
    return (vec / _norm(vec)).astype(vec.dtype)


def normalize_ip(vec: np.ndarray) -> None:
    vec /= _norm(vec)


def mat33_create_from_quaternion(quat: np.ndarray) -> np.ndarray:
    """Rotation matrix rows [m00 m01 m02; m10 m11 m12; m20 m21 m22] (functions.py:35-82)."""
    q = np.asarray(quat)
    dt = q.dtype
    if not np.isclose(_norm(q), 1.0):
        q = normalize(q)
    x, y, z, w = (q[0], q[1], q[2], q[3])
    xx, yy, zz, ww = x * x, y * y, z * z, w * w          # dtype of quat
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    inv = 1.0 / np.float64(xx + yy + zz + ww)            # int/float32 -> float64 under numba
    m = np.empty((3, 3), dtype=np.float64)
    m[0, 0] = np.float64(xx - yy - zz + ww) * inv
    m[1, 1] = np.float64(-xx + yy - zz + ww) * inv
    m[2, 2] = np.float64(-xx - yy + zz + ww) * inv
    m[1, 0] = 2.0 * np.float64(xy + zw) * inv
    m[0, 1] = 2.0 * np.float64(xy - zw) * inv
    m[2, 0] = 2.0 * np.float64(xz - yw) * inv
    m[0, 2] = 2.0 * np.float64(xz + yw) * inv
    m[2, 1] = 2.0 * np.float64(yz + xw) * inv
    m[1, 2] = 2.0 * np.float64(yz - xw) * inv
    return m.astype(dt)


def mat44_create_from_quaternion(quat: np.ndarray) -> np.ndarray:
    """float64 eye(4) with the (dtype-of-quat) rotation in [0:3,0:3] (functions.py:84-88)."""
    out = np.eye(4, dtype=np.float64)
    out[0:3, 0:3] = mat33_create_from_quaternion(quat)
    return out


def quaternion_create_from_axis_rotation(axis: np.ndarray, theta) -> np.ndarray:
    """[s*ax, s*ay, s*az, cos(theta/2)], s = sin(theta/2) in float64 (functions.py:92-110)."""
    axis = np.asarray(axis)
    dt = axis.dtype
    if not np.isclose(_norm(axis), 1.0):
        axis = normalize(axis)
    half = np.float64(theta) * 0.5
    s = np.sin(half)
    return np.array(
        [s * np.float64(axis[0]), s * np.float64(axis[1]), s * np.float64(axis[2]), np.cos(half)],
        dtype=dt,
    )


def multiply_quaternions(q1: np.ndarray, q2: np.ndarray) -> np.ndarray:
    """Hamilton product in the component order of functions.py:112-128, dtype of q1."""
    q1 = np.asarray(q1)
    q2 = np.asarray(q2)
    ax, ay, az, aw = q1
    bx, by, bz, bw = q2
    return np.array(
        [
            ax * bw + ay * bz - az * by + aw * bx,
            -ax * bz + ay * bw + az * bx + aw * by,
            ax * by - ay * bx + az * bw + aw * bz,
            -ax * bx - ay * by - az * bz + aw * bw,
        ],
        dtype=q1.dtype,
    )
