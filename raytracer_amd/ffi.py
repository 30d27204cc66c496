"""ctypes binding of include/rt_hip.h.

``RtLib()`` loads ``raytracer_amd/librt_hip.so`` (the hand-written gfx950 kernels) and
raises immediately if it is absent - there is NO CPU fallback in the product.  Tests may
pass another path (the CPU oracle exports the same C-ABI) to drive both with one binding.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

HIP_LIB = Path(__file__).resolve().parent / "librt_hip.so"

RT_ACCUM_MEAN = 0
RT_ACCUM_SUM = 1
RT_CULL_OFF, RT_CULL_ON, RT_CULL_BAND = 0, 1, 2


class RtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("samples", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("launches", C.c_uint64),
        ("sched", C.c_uint64 * 6),
        ("cull_audit", C.c_uint64 * 3),
    ]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["sched"] = list(self.sched)
        d["cull_audit"] = list(self.cull_audit)
        return d


class RtInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("accel", "n_slots", "n_records", "top_level_tree", "stack_entries", "stack_entries_lds",
                                         "lds_bytes", "workgroups_per_cu", "top_level_rebuilds", "n_roots", "root_table_variants")] + [("reserved", C.c_int32 * 5)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class RtError(RuntimeError):
    def __init__(self, code: int, where: str, text: str):
        super().__init__(f"{where} failed with code {code}: {text}")
        self.code = code


_vp = C.c_void_p
_SIGNATURES = {
    "rt_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "rt_destroy": (C.c_int, [_vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_backend_name": (C.c_char_p, []),
    "rt_set_scene": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp]),
    "rt_set_materials": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "rt_update_mesh_transform": (C.c_int, [_vp, C.c_int, _vp]),
    "rt_update_mesh_pose": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "rt_read_tris28": (C.c_int, [_vp, _vp]),
    "rt_read_nodes48": (C.c_int, [_vp, _vp]),
    "rt_bvh_build": (C.c_int, [_vp, C.c_int, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int, _vp, C.POINTER(C.c_int32)]),
    "rt_bvh_refit": (C.c_int, [_vp, C.c_int, C.c_int32, C.c_int32, _vp, C.c_int, _vp]),
    "rt_set_camera": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, C.c_float]),
    "rt_resize": (C.c_int, [_vp, C.c_int, C.c_int]),
    "rt_reset_accum": (C.c_int, [_vp]),
    "rt_bind_accum": (C.c_int, [_vp, _vp]),
    "rt_render": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]),
    "rt_render_static": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]),
    "rt_prepare": (C.c_int, [_vp]),
    "rt_set_max_cycles": (C.c_int, [_vp, C.c_uint32]),
    "rt_first_hit": (C.c_int, [_vp, C.c_uint32, _vp, _vp]),
    "rt_read_accum": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64)]),
    "rt_read_mean": (C.c_int, [_vp, _vp]),
    "rt_set_accum_passes": (C.c_int, [_vp, C.c_uint64]),
    "rt_allreduce_sum": (C.c_int, [_vp, _vp, C.c_uint64]),
    "rt_set_cull": (C.c_int, [_vp, C.c_int]),
    "rt_set_accel": (C.c_int, [_vp, C.c_int]),
    "rt_enable_stats": (C.c_int, [_vp, C.c_int]),
    "rt_reset_stats": (C.c_int, [_vp]),
    "rt_get_stats": (C.c_int, [_vp, C.POINTER(RtStats)]),
    "rt_get_info": (C.c_int, [_vp, C.POINTER(RtInfo)]),
    "rt_sync": (C.c_int, [_vp]),
    "rt_set_stream": (C.c_int, [_vp, _vp]),
    "rt_stream": (_vp, [_vp]),
}
ABI_SYMBOLS = tuple(_SIGNATURES)


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(_vp)


def _f32(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and out.shape != shape:
        raise ValueError(f"expected shape {shape}, got {out.shape}")
    return out


class RtLib:
    """A loaded C-ABI library."""

    def __init__(self, path: str | Path | None = None):
        path = Path(path) if path is not None else HIP_LIB
        if not path.exists():
            raise FileNotFoundError(
                f"{path} is missing: the HIP extension must be built first "
                f"(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
            )
        self.path = path
        self.dll = C.CDLL(str(path))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.dll, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        self.backend = self.dll.rt_backend_name().decode()

    def create(self, device: int = 0) -> "RtContext":
        return RtContext(self, device)

    # -- host-side BVH (no context, no GPU) -------------------------------------------------
    def bvh_build(self, pos: np.ndarray, first_node_id: int = 0, first_tri_id: int = 0, first_leaf_offset: int = 0,
                  node_dtype=None):
        """(nodes48, leaf_tri_ids) of one mesh, built natively with the reference's split rule.
        pos = (n, 3, 3) float32 corner positions."""
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 9)
        n = pos.shape[0]
        cap = 2 * n
        nodes = np.zeros(cap * 48, dtype=np.uint8)
        ids = np.zeros(n, dtype=np.int32)
        n_nodes = C.c_int32(0)
        rc = self.dll.rt_bvh_build(_ptr(pos), n, first_node_id, first_tri_id, first_leaf_offset, _ptr(nodes), cap, _ptr(ids),
                                   C.byref(n_nodes))
        if rc != 0:
            raise RtError(rc, "rt_bvh_build", "bad arguments" if rc == -1 else "node capacity")
        nodes = nodes[: n_nodes.value * 48].copy()
        return (nodes.view(node_dtype) if node_dtype is not None else nodes), ids

    def bvh_refit(self, pos: np.ndarray, nodes: np.ndarray, leaf_ids: np.ndarray, first_tri_id: int = 0,
                  first_leaf_offset: int = 0) -> None:
        """Recompute the boxes of `nodes` (modified in place) from the current positions."""
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 9)
        assert nodes.flags.c_contiguous and nodes.nbytes % 48 == 0
        leaf_ids = np.ascontiguousarray(leaf_ids, dtype=np.int32)
        rc = self.dll.rt_bvh_refit(_ptr(pos), pos.shape[0], first_tri_id, first_leaf_offset, _ptr(nodes), nodes.nbytes // 48,
                                   _ptr(leaf_ids))
        if rc != 0:
            raise RtError(rc, "rt_bvh_refit", "inconsistent arguments")


class RtContext:
    """One rendering context (one GPU)."""

    def __init__(self, lib: RtLib, device: int = 0):
        self.lib = lib
        self._h = _vp()
        rc = lib.dll.rt_create(device, C.byref(self._h))
        if rc != 0:
            raise RtError(rc, "rt_create", "could not create context")
        self.width = self.height = 0
        self.n_tris = self.n_nodes = 0

    # -- plumbing ----------------------------------------------------------------------
    def _ck(self, rc: int, where: str) -> None:
        if rc != 0:
            raise RtError(rc, where, (self.lib.dll.rt_last_error(self._h) or b"").decode())

    def close(self) -> None:
        if self._h:
            self.lib.dll.rt_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene ---------------------------------------------------------------------------
    def set_scene(self, tris28, meshes128, nodes48, leaf_tri_ids, spheres96=None, atmosphere80=None) -> None:
        tris = _f32(tris28).reshape(-1, 28)
        meshes = np.ascontiguousarray(meshes128)
        nodes = np.ascontiguousarray(nodes48)
        ids = np.ascontiguousarray(leaf_tri_ids, dtype=np.int32)
        sph = None if spheres96 is None or len(spheres96) == 0 else np.ascontiguousarray(spheres96)
        atm = np.ascontiguousarray(atmosphere80)
        assert meshes.dtype.itemsize == 128 and nodes.dtype.itemsize == 48 and atm.nbytes == 80
        self._ck(
            self.lib.dll.rt_set_scene(
                self._h, _ptr(tris), tris.shape[0], _ptr(meshes), meshes.shape[0], _ptr(nodes), nodes.shape[0],
                _ptr(ids), ids.shape[0], _ptr(sph), 0 if sph is None else sph.shape[0], _ptr(atm),
            ),
            "rt_set_scene",
        )
        self.n_tris, self.n_nodes = tris.shape[0], nodes.shape[0]

    def set_materials(self, meshes128, atmosphere80=None) -> None:
        meshes = np.ascontiguousarray(meshes128)
        atm = None if atmosphere80 is None else np.ascontiguousarray(atmosphere80)
        self._ck(self.lib.dll.rt_set_materials(self._h, _ptr(meshes), meshes.shape[0], _ptr(atm)), "rt_set_materials")

    def update_mesh_transform(self, mesh: int, T44) -> None:
        t = _f32(T44).reshape(16)
        self._ck(self.lib.dll.rt_update_mesh_transform(self._h, mesh, _ptr(t)), "rt_update_mesh_transform")

    def update_mesh_pose(self, mesh: int, pos, quat) -> None:
        p, q = _f32(pos, (3,)), _f32(quat, (4,))
        self._ck(self.lib.dll.rt_update_mesh_pose(self._h, mesh, _ptr(p), _ptr(q)), "rt_update_mesh_pose")

    def read_tris28(self) -> np.ndarray:
        out = np.empty((self.n_tris, 28), dtype=np.float32)
        self._ck(self.lib.dll.rt_read_tris28(self._h, _ptr(out)), "rt_read_tris28")
        return out

    def read_nodes48(self, dtype) -> np.ndarray:
        out = np.empty(self.n_nodes, dtype=dtype)
        self._ck(self.lib.dll.rt_read_nodes48(self._h, _ptr(out)), "rt_read_nodes48")
        return out

    # -- camera / image ------------------------------------------------------------------------
    def set_camera(self, view_params, local_to_world16, cam_pos, dof: float, aa: float) -> None:
        vp, m, p = _f32(view_params, (3,)), _f32(local_to_world16).reshape(16), _f32(cam_pos, (3,))
        self._ck(self.lib.dll.rt_set_camera(self._h, _ptr(vp), _ptr(m), _ptr(p), float(dof), float(aa)), "rt_set_camera")

    def resize(self, width: int, height: int) -> None:
        self._ck(self.lib.dll.rt_resize(self._h, width, height), "rt_resize")
        self.width, self.height = width, height

    def reset_accum(self) -> None:
        self._ck(self.lib.dll.rt_reset_accum(self._h), "rt_reset_accum")

    def bind_accum(self, ptr: int | None) -> None:
        self._ck(self.lib.dll.rt_bind_accum(self._h, _vp(ptr) if ptr else None), "rt_bind_accum")

    def set_accum_passes(self, n: int) -> None:
        self._ck(self.lib.dll.rt_set_accum_passes(self._h, n), "rt_set_accum_passes")

    def allreduce_sum(self, nccl_comm: int, total_passes: int) -> None:
        """ncclAllReduce(sum) of the accumulator over an ncclComm_t handle, on the context's stream (rt_hip.h)."""
        self._ck(self.lib.dll.rt_allreduce_sum(self._h, _vp(nccl_comm), total_passes), "rt_allreduce_sum")

    # -- work ------------------------------------------------------------------------------------
    def render(self, first_pass: int, n_passes: int, pass_stride: int = 1, rays_per_pixel: int = 1,
               max_bounces: int = 1, flags: int = RT_ACCUM_MEAN) -> None:
        self._ck(
            self.lib.dll.rt_render(self._h, first_pass, n_passes, pass_stride, rays_per_pixel, max_bounces, flags),
            "rt_render",
        )

    def render_static(self, first_pass: int, n_passes: int, pass_stride: int = 1, rays_per_pixel: int = 1,
                      max_bounces: int = 1, flags: int = RT_ACCUM_MEAN) -> None:
        """rt_render without any allocation: scratch in the library's static device array (rt_hip.h)."""
        self._ck(
            self.lib.dll.rt_render_static(self._h, first_pass, n_passes, pass_stride, rays_per_pixel, max_bounces, flags),
            "rt_render_static",
        )

    def prepare(self) -> None:
        """Apply pending pose / material updates on the device now (rt_render does it implicitly, rt_render_static refuses to)."""
        self._ck(self.lib.dll.rt_prepare(self._h), "rt_prepare")

    def set_max_cycles(self, max_cycles: int) -> None:
        """MAX_CYCLES uniform (application.py:997): passes with a larger index leave the image unchanged."""
        self._ck(self.lib.dll.rt_set_max_cycles(self._h, int(max_cycles) & 0xFFFFFFFF), "rt_set_max_cycles")

    def first_hit(self, pass_index: int = 1) -> tuple[np.ndarray, np.ndarray]:
        ids = np.empty((self.height, self.width), dtype=np.int32)
        ts = np.empty((self.height, self.width), dtype=np.float32)
        self._ck(self.lib.dll.rt_first_hit(self._h, pass_index, _ptr(ids), _ptr(ts)), "rt_first_hit")
        return ids, ts

    def read_accum(self) -> tuple[np.ndarray, int]:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        n = C.c_uint64(0)
        self._ck(self.lib.dll.rt_read_accum(self._h, _ptr(out), C.byref(n)), "rt_read_accum")
        return out, int(n.value)

    def read_mean(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._ck(self.lib.dll.rt_read_mean(self._h, _ptr(out)), "rt_read_mean")
        return out

    # -- counters ------------------------------------------------------------------------------------
    def set_accel(self, sah: bool) -> None:
        """Structure the NEXT set_scene builds: False = the reference's BVH (parity), True = SAH."""
        self._ck(self.lib.dll.rt_set_accel(self._h, 1 if sah else 0), "rt_set_accel")

    def set_cull(self, mode) -> None:
        """False / 0 = exact box set, True / 1 = both culling rules, 2 (RT_CULL_BAND) = the distance band alone."""
        self._ck(self.lib.dll.rt_set_cull(self._h, int(mode)), "rt_set_cull")

    def enable_stats(self, on: bool = True) -> None:
        self._ck(self.lib.dll.rt_enable_stats(self._h, int(on)), "rt_enable_stats")

    def reset_stats(self) -> None:
        self._ck(self.lib.dll.rt_reset_stats(self._h), "rt_reset_stats")

    def stats(self) -> dict:
        s = RtStats()
        self._ck(self.lib.dll.rt_get_stats(self._h, C.byref(s)), "rt_get_stats")
        return s.as_dict()

    def info(self) -> dict:
        """What rt_set_scene built and how the most recent trace launch was laid out (rt_info_t)."""
        i = RtInfo()
        self._ck(self.lib.dll.rt_get_info(self._h, C.byref(i)), "rt_get_info")
        return i.as_dict()

    def sync(self) -> None:
        self._ck(self.lib.dll.rt_sync(self._h), "rt_sync")

    def set_stream(self, hip_stream: int | None) -> None:
        """hip_stream = a hipStream_t handle; 0 = HIP's null stream (torch's default stream);
        None = back to the context's own stream."""
        h = _vp(-1) if hip_stream is None else _vp(hip_stream)     # RT_STREAM_OWN = (void*)-1
        self._ck(self.lib.dll.rt_set_stream(self._h, h), "rt_set_stream")

    def stream(self) -> int:
        return int(self.lib.dll.rt_stream(self._h) or 0)

    # -- convenience: a whole Scene object ----------------------------------------------------------
    def load_scene(self, scene, rest_pose: bool = True) -> None:
        """Upload a ``scripts.classes.Scene``: rest-pose geometry + per-mesh poses applied on
        the device (rest_pose=True), or host-posed geometry as the reference uploads it."""
        from .scripts import functions as F

        if not rest_pose:
            F.pose_dirty_meshes_on_host(scene)
        b = F.export_scene_buffers(scene, rest_pose=rest_pose)
        self.set_scene(b["tris28"], b["meshes128"], b["nodes48"], b["leaf_tri_ids"], b["spheres96"], b["atmosphere80"])
        if rest_pose:
            for mesh in scene.meshes:
                self.update_mesh_pose(mesh.id, mesh.csys.pos, mesh.csys.quat)
                mesh.unflag_for_mesh_update()

    def update_scene(self, scene) -> None:
        """Per-frame update of an uploaded scene (what the reference re-uploads each frame,
        application.py:939-976): poses of the meshes flagged for update, every material."""
        from .scripts.classes import MESH_DTYPE

        for mesh in scene.meshes:
            if mesh.is_awaiting_mesh_update:
                self.update_mesh_pose(mesh.id, mesh.csys.pos, mesh.csys.quat)
                mesh.unflag_for_mesh_update()
        recs = np.zeros(len(scene.meshes), dtype=MESH_DTYPE)
        for i, m in enumerate(scene.meshes):
            recs[i]["index"] = m.id
            recs[i]["material"] = m.material.as_record()
        self.set_materials(recs, scene.atmosphere_material.as_record().reshape(1))

    def load_camera(self, cam) -> None:
        from .scripts import functions as F

        u = F.camera_uniforms(cam)
        self.set_camera(u["view_params"], u["local_to_world16"], u["cam_pos"], u["dof"], u["aa"])
