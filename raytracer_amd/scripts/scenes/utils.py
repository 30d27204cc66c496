"""Scene-building utilities (reference ``scripts/scenes/utils.py``): OBJ import without
trimesh, registry reset, smoothstep/blend.

OBJ semantics fixed by this build ("parity unpinned vs trimesh" - trimesh is
a third-party dependency of the reference, un-pinned and absent here): every face corner
takes the file's ``v`` and ``vn`` for that corner; polygons are fan-triangulated
(0,1,2),(0,2,3)...; face order = file order; values are cast to float32.  A file without
``vn`` gets the normalised geometric face normal on all three corners.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from ..numba_utils import classes as _nb

ASSET_ROOT = Path(__file__).resolve().parents[3] / "assets"


def resolve_asset(f) -> Path:
    """Reference scenes use paths relative to the repo root ("objects/monkey_blend/...")."""
    p = Path(f)
    if p.exists():
        return p
    q = ASSET_ROOT / p
    if q.exists():
        return q
    raise FileNotFoundError(f"{f} (also tried {q})")


def read_obj(f) -> tuple[np.ndarray, np.ndarray]:
    """-> (pos, nrm) each (n_tris, 3, 3) float32."""
    vs: list[tuple[float, float, float]] = []
    vns: list[tuple[float, float, float]] = []
    corners_v: list[int] = []
    corners_n: list[int] = []
    with open(resolve_asset(f), "r") as fh:
        for line in fh:
            if line.startswith("v "):
                a = line.split()
                vs.append((float(a[1]), float(a[2]), float(a[3])))
            elif line.startswith("vn "):
                a = line.split()
                vns.append((float(a[1]), float(a[2]), float(a[3])))
            elif line.startswith("f "):
                vi, ni = [], []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    vi.append(i - 1 if i > 0 else len(vs) + i)
                    if len(parts) > 2 and parts[2]:
                        j = int(parts[2])
                        ni.append(j - 1 if j > 0 else len(vns) + j)
                    else:
                        ni.append(-1)
                for k in range(1, len(vi) - 1):
                    corners_v += (vi[0], vi[k], vi[k + 1])
                    corners_n += (ni[0], ni[k], ni[k + 1])
    v = np.asarray(vs, dtype=np.float64).astype(np.float32)
    pos = v[np.asarray(corners_v, dtype=np.int64)].reshape(-1, 3, 3)
    cn = np.asarray(corners_n, dtype=np.int64).reshape(-1, 3)
    if len(vns) and (cn >= 0).all():
        vn = np.asarray(vns, dtype=np.float64).astype(np.float32)
        nrm = vn[cn]
    else:
        e1 = pos[:, 1] - pos[:, 0]
        e2 = pos[:, 2] - pos[:, 0]
        g = np.cross(e1, e2)
        g = g / np.linalg.norm(g, axis=1, keepdims=True)
        nrm = np.repeat(g[:, None, :], 3, axis=1).astype(np.float32)
    return np.ascontiguousarray(pos), np.ascontiguousarray(nrm)


def get_triangles_from_obj(f, mesh_idx):
    """Load, give the triangles the next global ids, append to the registry (utils.py:11-30)."""
    pos, nrm = read_obj(f)
    if pos.shape[0] == 0:
        raise Exception(f"no triangles in {f}")
    start = len(_nb.get_all_triangles_arr())
    tris = _nb.TriangleSet(pos, nrm, mesh_idx, start)
    _nb.add_to_all_triangles(tris)
    return tris


def reset_global_datas() -> None:
    from ..classes import Mesh
    from . import chunker

    _nb.reset_all_triangles()
    Mesh.reset()
    chunker.BVHGraph.reset()
    chunker.BVHParentNode.reset()


def subdivide_triangles(pos: np.ndarray, nrm: np.ndarray, levels: int):
    """Pure 1 -> 4 midpoint split, `levels` times (BASELINE.json configs[3], SURVEY.md 8(d)
    config 4): new corners are float32 edge midpoints (projected nowhere), their normals the
    renormalised sum of the edge's two corner normals.  Children of a triangle stay together,
    in the order (A,ab,ca) (ab,B,bc) (ca,bc,C) (ab,bc,ca).  Deterministic synthetic data."""
    pos = np.asarray(pos, dtype=np.float32)
    nrm = np.asarray(nrm, dtype=np.float32)

    def mid_n(a, b):
        v = a + b
        ln = np.sqrt((v * v).sum(axis=-1, keepdims=True)).astype(np.float32)
        return np.where(ln > 0, v / np.where(ln > 0, ln, 1), a).astype(np.float32)

    for _ in range(levels):
        a, b, c = pos[:, 0], pos[:, 1], pos[:, 2]
        na, nb, nc = nrm[:, 0], nrm[:, 1], nrm[:, 2]
        h = np.float32(0.5)
        ab, bc, ca = (a + b) * h, (b + c) * h, (c + a) * h
        nab, nbc, nca = mid_n(na, nb), mid_n(nb, nc), mid_n(nc, na)
        pos = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1)
        nrm = np.stack([np.stack([na, nab, nca], 1), np.stack([nab, nb, nbc], 1), np.stack([nca, nbc, nc], 1),
                        np.stack([nab, nbc, nca], 1)], 1)
        pos = np.ascontiguousarray(pos.reshape(-1, 3, 3), dtype=np.float32)
        nrm = np.ascontiguousarray(nrm.reshape(-1, 3, 3), dtype=np.float32)
    return pos, nrm


def smoothstep(edge0, edge1, x):
    t = np.clip((x - edge0) / (edge1 - edge0), 0.0, 1.0)
    return t * t * (3 - 2 * t)


def blend(factor, a, b):
    return (1 - factor) * a + factor * b
