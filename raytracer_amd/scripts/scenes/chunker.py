"""Per-mesh BVH construction and refit with the reference's split rule
(``scripts/scenes/chunker.py``), index-based and vectorised.

Rule restated: a node holds an ordered list of its mesh's triangles; it
stops at <= 5 triangles or depth 20; otherwise the longest AABB axis is halved and a
triangle goes LEFT iff its centroid (float32 mean of its three corners, frozen at node
construction) lies inside the lower half-box inclusive on all six faces, else RIGHT; if
either side would be empty the node stays an (oversized) leaf.  Children are created
left-then-right *before* either subtree is descended, which fixes the global node ids.
Refit recomputes every node's AABB from the current positions of its own triangles;
topology never changes.

Nodes keep an index array into their mesh's ``TriangleSet`` rather than a Python list of
triangle objects, so a node's vertex block is one fancy-index gather.
"""
from __future__ import annotations

import numpy as np

NODE_DTYPE = np.dtype(
    [
        ("aabbMin", "<f4", 4),       # @0  (w = 0)
        ("aabbMax", "<f4", 3),       # @16
        ("id", "<i4"),               # @28
        ("childId1", "<i4"),         # @32
        ("childId2", "<i4"),         # @36
        ("childObjOffset", "<i4"),   # @40
        ("childObjCount", "<i4"),    # @44
    ]
)
assert NODE_DTYPE.itemsize == 48

MAX_LEAF_TRIS = 5
MAX_DEPTH = 20


def point_in_aabb(point, aabb) -> bool:
    return bool(np.all(point >= aabb[0]) and np.all(point <= aabb[1]))


def chunk_mesh_bvh(mesh, lib=None) -> None:
    """Build ``mesh.bvh_graph`` (chunker.py:17-25).  With ``lib`` (a loaded C-ABI library,
    ``raytracer_amd.ffi.RtLib``) the graph is built natively by ``rt_bvh_build`` - same rule,
    same bytes (tests/test_bvh_native.py) - which is what makes 10^6-triangle meshes usable."""
    if not len(mesh.triangles):
        print("Warning, no triangles detected in mesh, returning without mesh!")
        raise Exception
    if lib is not None:
        mesh.bvh_graph = NativeBVHGraph(mesh, lib)
        return
    graph = BVHGraph(mesh=mesh)
    root = BVHParentNode(graph=graph, tris=np.arange(len(mesh.triangles), dtype=np.int64), depth=0)
    root.split_recursively(max_depth=MAX_DEPTH)
    mesh.bvh_graph = graph


class BVHGraph:
    BVH_TRI_ID_LIST_GLOBAL: list[int] = []
    ALL: list["BVHGraph"] = []

    N_NODES_GLOBAL = 0       # node ids are global over all graphs, in creation order

    def __init__(self, mesh) -> None:
        self.id = len(BVHGraph.ALL)
        BVHGraph.ALL.append(self)
        self.mesh = mesh
        self._nodes: list[BVHParentNode] = []
        self.leaf_nodes: list[BVHParentNode] | None = None
        self.is_awaiting_mesh_update = False

    @property
    def nodes(self) -> list["BVHParentNode"]:
        return self._nodes

    def flag_for_mesh_update(self) -> None:
        self.is_awaiting_mesh_update = True

    def unflag_for_mesh_update(self) -> None:
        self.is_awaiting_mesh_update = False

    @property
    def first_node_id(self) -> int:
        return self._nodes[0].id

    def register_node(self, node: "BVHParentNode") -> None:
        self._nodes.append(node)

    def update_leaf_nodes_and_tri_ids(self) -> None:
        self.leaf_nodes = [n for n in self._nodes if n.is_leaf()]

    @staticmethod
    def register_all_and_update_node_tri_ids() -> None:
        """Flat leaf triangle-id list: graphs in creation order, leaves in creation order
        (chunker.py:79-86)."""
        parts: list[np.ndarray] = []
        at = 0
        for g in BVHGraph.ALL:
            at = g._register_leaves(at, parts)
        BVHGraph.BVH_TRI_ID_LIST_GLOBAL = np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)

    def _register_leaves(self, at: int, parts: list) -> int:
        self.update_leaf_nodes_and_tri_ids()
        for node in self.leaf_nodes:
            node.tris_start_offset = at
            parts.append(node.tri_ids)
            at += node.tris_count
        return at

    def node_records(self) -> np.ndarray:
        """std140 image of this graph's nodes, creation order."""
        out = np.zeros(len(self._nodes), dtype=NODE_DTYPE)
        for i, n in enumerate(self._nodes):
            out[i] = n.as_record()
        return out

    @staticmethod
    def all_node_records() -> np.ndarray:
        recs = [g.node_records() for g in BVHGraph.ALL]
        return np.concatenate(recs) if recs else np.zeros(0, dtype=NODE_DTYPE)

    @staticmethod
    def reset() -> None:
        BVHGraph.BVH_TRI_ID_LIST_GLOBAL = []
        BVHGraph.ALL = []
        BVHGraph.N_NODES_GLOBAL = 0

    def update_graph_node_aabbs_for_changed_triangles(self, force: bool = False) -> None:
        if (not force) and (not self.is_awaiting_mesh_update):
            return
        for node in self._nodes:
            node.update_for_modified_triangles()
        self.unflag_for_mesh_update()


class NativeBVHGraph(BVHGraph):
    """A graph held as the arrays ``rt_bvh_build`` returns (no per-node Python objects)."""

    def __init__(self, mesh, lib) -> None:
        super().__init__(mesh)
        self.lib = lib
        ts = mesh.triangles
        self.first_tri_id = int(ts.ids[0])
        assert np.array_equal(ts.ids, np.arange(self.first_tri_id, self.first_tri_id + len(ts), dtype=ts.ids.dtype))
        self.nodes48, self.leaf_ids = lib.bvh_build(ts.pos, BVHGraph.N_NODES_GLOBAL, self.first_tri_id, 0, NODE_DTYPE)
        BVHGraph.N_NODES_GLOBAL += len(self.nodes48)
        self.leaf_base = 0

    @property
    def nodes(self):
        raise AttributeError("a native graph has no node objects; use node_records()")

    @property
    def first_node_id(self) -> int:
        return int(self.nodes48["id"][0])

    def _register_leaves(self, at: int, parts: list) -> int:
        leaf = self.nodes48["childObjOffset"] >= 0
        self.nodes48["childObjOffset"][leaf] += at - self.leaf_base
        self.leaf_base = at
        parts.append(self.leaf_ids)
        return at + len(self.leaf_ids)

    def node_records(self) -> np.ndarray:
        return self.nodes48

    def update_graph_node_aabbs_for_changed_triangles(self, force: bool = False) -> None:
        if (not force) and (not self.is_awaiting_mesh_update):
            return
        self.lib.bvh_refit(self.mesh.triangles.pos, self.nodes48, self.leaf_ids, self.first_tri_id, self.leaf_base)
        self.unflag_for_mesh_update()


class BVHParentNode:
    ALL: list["BVHParentNode"] = []

    def __init__(self, graph: BVHGraph, tris: np.ndarray, depth: int) -> None:
        self.graph = graph
        self.id = BVHGraph.N_NODES_GLOBAL
        BVHGraph.N_NODES_GLOBAL += 1
        BVHParentNode.ALL.append(self)
        self.depth = depth
        self.child_left: BVHParentNode | None = None
        self.child_right: BVHParentNode | None = None
        self.tri_idx = np.asarray(tris, dtype=np.int64)   # local indices, parent order kept
        self.tris_count = int(self.tri_idx.shape[0])
        self.tris_start_offset: int | None = None
        self.vertices: np.ndarray | None = None
        self.aabb: np.ndarray | None = None
        self._update_vertices()
        # centroids are frozen here (the reference never refreshes them, chunker.py:135-138)
        self.centroids = np.mean(self.vertices, axis=1) if self.tris_count else np.zeros((0, 3), np.float32)
        self._update_aabb()
        graph.register_node(self)

    @classmethod
    def reset(cls) -> None:
        cls.ALL = []

    # -- reference-compatible views ------------------------------------------------------
    @property
    def tris(self):
        ts = self.graph.mesh.triangles
        return [ts[int(i)] for i in self.tri_idx]

    @property
    def tri_ids(self) -> np.ndarray:
        return self.graph.mesh.triangles.ids[self.tri_idx].astype(np.int32)

    @property
    def child_left_id(self) -> int:
        return -1 if self.child_left is None else self.child_left.id

    @property
    def child_right_id(self) -> int:
        return -1 if self.child_right is None else self.child_right.id

    def is_leaf(self) -> bool:
        return self.child_left is None and self.child_right is None

    # -- build -----------------------------------------------------------------------------
    def split_recursively(self, max_depth: int) -> None:
        if self.tris_count <= MAX_LEAF_TRIS or self.depth == max_depth:
            return
        self.split()
        if self.child_left is not None:
            self.child_left.split_recursively(max_depth=max_depth)
        if self.child_right is not None:
            self.child_right.split_recursively(max_depth=max_depth)

    def split(self) -> None:
        lo, hi = self.aabb[0], self.aabb[1]
        axis = int(np.argmax(np.abs(lo - hi)))
        half = np.zeros(3, dtype=np.float32)
        half[axis] = 0.5
        left_hi = hi - (hi - lo) * half
        inside = ((self.centroids >= lo) & (self.centroids <= left_hi)).all(axis=1)
        n_left = int(inside.sum())
        if n_left == 0 or n_left == self.tris_count:
            return
        self.child_left = BVHParentNode(self.graph, self.tri_idx[inside], self.depth + 1)
        self.child_right = BVHParentNode(self.graph, self.tri_idx[~inside], self.depth + 1)

    # -- refit -----------------------------------------------------------------------------
    def update_for_modified_triangles(self) -> None:
        self._update_vertices()
        self._update_aabb()

    def _update_vertices(self) -> None:
        self.vertices = self.graph.mesh.triangles.pos[self.tri_idx]   # (n,3,3) float32

    def _update_aabb(self) -> None:
        if not self.tris_count:
            self.aabb = np.zeros((2, 3), dtype=np.float32)
            return
        pts = self.vertices.reshape(-1, 3)
        self.aabb = np.stack([pts.min(axis=0), pts.max(axis=0)])

    # -- std140 image ------------------------------------------------------------------------
    def as_record(self) -> np.ndarray:
        r = np.zeros((), dtype=NODE_DTYPE)
        r["aabbMin"][:3] = self.aabb[0]
        r["aabbMax"] = self.aabb[1]
        r["id"] = self.id
        r["childId1"] = self.child_left_id
        r["childId2"] = self.child_right_id
        r["childObjOffset"] = -1 if self.tris_start_offset is None else self.tris_start_offset
        r["childObjCount"] = self.tris_count
        return r

    def tobytes(self) -> bytes:
        return self.as_record().tobytes()
