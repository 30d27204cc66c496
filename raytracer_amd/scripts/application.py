"""Headless application shell and the HIP-backed render program (reference:
``scripts/application.py``).

Kept from the reference, with the same names and call order, is exactly what sits on the
render path: ``set_scene``, ``Application`` (``register_program`` / ``start`` / ``run`` /
``animate_frame``), the ``ProgramABC`` plugin interface (application.py:475-611) and the
``RayTracerDynamic`` program (``name = "raytracer"``, application.py:820-1223) whose methods
now forward to the C-ABI of ``librt_hip.so`` instead of moderngl:

=====================================  ==================================================
reference (moderngl)                   this build (include/rt_hip.h)
=====================================  ==================================================
context/program/VAO creation           ``rt_create``
texA/texB ping-pong, ``previousFrame``  one accumulator in HBM (``rt_resize``)
SSBO/UBO uploads every frame           ``rt_set_scene`` once, then ``rt_update_mesh_pose``
                                       for flagged meshes + ``rt_set_materials``
numba transform + chunker refit        device kernels behind ``rt_update_mesh_pose``
uniform writes                         ``rt_set_camera`` and ``rt_render`` arguments
``vao.render`` per pass and chunk      ``rt_render(first_pass, n_passes, ...)``
``fbo.read``                           ``rt_read_mean``
=====================================  ==================================================

Out of scope: the pygame window, key/mouse camera, TAB toggle,
watchdog hot reload and the unlit preview renderer.  ``handle_event`` exists and ignores
events, as the reference's own does (application.py:1225-1238).
"""
from __future__ import annotations

import datetime
import threading
from abc import ABC
from pathlib import Path

import numpy as np

from .. import ffi
from . import functions
from .classes import MESH_DTYPE, Camera, Mesh, Scene, pack_records

SCENE = None
HEADLESS = True


def set_scene(mod) -> None:
    """Select the scene module (must expose ``scene`` and ``get_frame_number``)."""
    global SCENE
    SCENE = mod


def compute_chunks(width, height, chunksx, chunksy):
    """Scissor rectangles (x, y, w, h), y flipped to a bottom-left origin (application.py:99-113).
    The HIP path does not need them (launch length is bounded inside ``rt_render``)."""
    cw, ch = width // chunksx, height // chunksy
    rects = []
    for i in range(chunksx):
        for j in range(chunksy):
            x = i * cw
            rects.append((x, height - (j + 1) * ch, cw if i < chunksx - 1 else width - x,
                          ch if j < chunksy - 1 else height - j * ch))
    return np.array(rects, dtype=np.int32)


class Application:
    """Owns the scene, the registered programs and the frame loop (application.py:116-394)."""

    def __init__(self, render_height: int | None = None, max_frames: int | None = None, program=None) -> None:
        factor = 4
        self.DYNAMIC_RENDER_FRAMERATE = 24
        self.RENDER_HEIGHT = render_height if render_height is not None else 1080 // factor
        self.WINDOW_HEIGHT = max(540, self.RENDER_HEIGHT)
        self.ASPECT_RATIO = 16 / 9
        self.MAX_CYCLES = 2048 * 2048
        self.CHUNKSX = 1
        self.CHUNKSY = 1
        self.display_program: ProgramABC | None = None
        self.programs: dict[str, ProgramABC] = {}
        self.scenes: dict[str, Scene] = {}
        self.running = False
        self.max_frames = max_frames            # headless budget replacing `while self.running` forever
        if SCENE is None:
            raise RuntimeError("call set_scene(module) before creating the Application")
        self.display_scene: Scene = SCENE.scene
        self.is_animating = False
        self.animation_clock = 0
        self.dt = 1 / self.DYNAMIC_RENDER_FRAMERATE
        self.frame_counter = 0
        self.is_requesting_exit = threading.Event()
        self.register_program(
            program if program is not None else
            RayTracerDynamic(app=self, file_fragment_shader="shaders/raytracer.fs.glsl",
                             file_vertex_shader="shaders/raytracer.vs.glsl",
                             width=self.RENDER_WIDTH, height=self.RENDER_HEIGHT)
        )
        self.display_program = self.programs["raytracer"]

    @property
    def RENDER_WIDTH(self) -> int:
        return int(self.RENDER_HEIGHT * self.ASPECT_RATIO)

    @property
    def WINDOW_WIDTH(self) -> int:
        return int(self.WINDOW_HEIGHT * self.ASPECT_RATIO)

    def register_program(self, program: "ProgramABC") -> None:
        if program.name in self.programs:
            print("Warning, program already regiested")
        self.programs[program.name] = program
        if len(self.programs) == 1:
            self.display_program = program
        else:
            self.display_program = self.programs.get("default", self.display_program)

    def start(self) -> None:
        self.running = True
        self.run()

    def end(self) -> None:
        self.running = False

    def run(self) -> None:
        self.display_program.configure_program(self.display_scene)
        self.is_requesting_exit.clear()
        frames = 0
        while self.running:
            self.display_program.calculate_frame(self.display_scene)
            frames += 1
            if self.max_frames is not None and frames >= self.max_frames:
                self.running = False

    def animate_next(self) -> None:
        if self.is_animating:
            self.display_scene.animate(time=self.animation_clock)
            self.animation_clock = self.animation_clock + self.dt

    def animate_frame(self, i) -> None:
        if self.is_animating:
            self.display_scene.animate_frame(frame=i)
            self.display_program.configure_buffers_and_uniforms(self.display_scene)


def write_pfm(path, img_bottom_up: np.ndarray) -> None:
    """Portable float map, RGB, little-endian (negative scale), rows bottom-to-top."""
    img = np.ascontiguousarray(img_bottom_up, dtype="<f4")
    h, w, c = img.shape
    assert c == 3
    with open(path, "wb") as f:
        f.write(f"PF\n{w} {h}\n-1.0\n".encode("ascii"))
        f.write(img.tobytes())


def read_pfm(path) -> np.ndarray:
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3)


def write_exr(path, img_top_down: np.ndarray) -> None:
    """OpenEXR 2 scanline file, uncompressed, three FLOAT channels (stored B, G, R: the format wants
    them alphabetical), rows top-to-bottom - what the reference's ``imageio.imwrite(..., format="exr")`` of
    the flipped float32 frame holds (application.py:1117), written without imageio."""
    import struct

    img = np.ascontiguousarray(img_top_down, dtype="<f4")
    h, w, c = img.shape
    assert c == 3

    def attr(name: str, typ: str, value: bytes) -> bytes:
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value

    chlist = b"".join(n + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    window = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (struct.pack("<ii", 20000630, 2)
              + attr("channels", "chlist", chlist)
              + attr("compression", "compression", b"\0")
              + attr("dataWindow", "box2i", window)
              + attr("displayWindow", "box2i", window)
              + attr("lineOrder", "lineOrder", b"\0")
              + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
              + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
              + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
              + b"\0")
    row_bytes = 3 * w * 4
    first = len(header) + 8 * h
    offsets = np.arange(h, dtype="<u8") * (8 + row_bytes) + first
    rows = np.empty((h, 8 + row_bytes), dtype=np.uint8)
    rows[:, 0:4] = np.arange(h, dtype="<i4").view(np.uint8).reshape(h, 4)
    rows[:, 4:8] = np.frombuffer(struct.pack("<i", row_bytes), dtype=np.uint8)
    # per scanline: all B, then all G, then all R
    rows[:, 8:] = np.ascontiguousarray(img[:, :, ::-1].transpose(0, 2, 1)).view(np.uint8).reshape(h, row_bytes)
    with open(path, "wb") as f:
        f.write(header)
        f.write(offsets.tobytes())
        f.write(rows.tobytes())


def read_exr(path) -> np.ndarray:
    """Reader for the files ``write_exr`` makes (uncompressed scanline FLOAT B/G/R) -> (h, w, 3) RGB, top-down."""
    import struct

    data = open(path, "rb").read()
    assert struct.unpack_from("<ii", data, 0) == (20000630, 2)
    pos, attrs = 8, {}
    while data[pos] != 0:
        end = data.index(b"\0", pos)
        name = data[pos:end].decode()
        tend = data.index(b"\0", end + 1)
        size = struct.unpack_from("<i", data, tend + 1)[0]
        attrs[name] = data[tend + 5:tend + 5 + size]
        pos = tend + 5 + size
    pos += 1
    assert attrs["compression"] == b"\0" and attrs["lineOrder"] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offsets = np.frombuffer(data, dtype="<u8", count=h, offset=pos)
    out = np.empty((h, w, 3), dtype=np.float32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", data, int(off))
        assert size == 3 * w * 4
        bgr = np.frombuffer(data, dtype="<f4", count=3 * w, offset=int(off) + 8).reshape(3, w)
        out[y - y0] = bgr[::-1].T
    return out


def tone_map(color):
    """Reinhard operator of the reference's export helpers (application.py:1463-1465; the same expression sits
    commented out in shaders/toScreen.fs.glsl:18-21 and raytracer.fs.glsl:919-922).  color: H x W x 3 or 4."""
    return color / (color + 1.0)


def gamma_correct(color, gamma=2.2):
    """application.py:1467-1468 (toScreen.fs.glsl:23-26).  Nothing on the live path applies either of the two
    (application.py:1104-1117 clips the raw mean); they are hooks for the caller."""
    return np.power(color, 1.0 / gamma)


def to_display(hdr, gamma=2.2):
    """What the reference's commented-out display pass would show (toScreen.fs.glsl:28-38):
    tone-mapped, gamma-corrected, 8-bit."""
    with np.errstate(invalid="ignore"):
        t = tone_map(np.maximum(hdr, 0.0))
    t = np.where(np.isnan(t), 1.0, t)                 # inf / (inf + 1): infinite radiance shows as white
    return (np.clip(gamma_correct(t, gamma), 0.0, 1.0) * 255).astype(np.uint8)


class ProgramABC(ABC):
    """Plugin interface of a render program (application.py:475-611)."""

    app: Application
    name: str
    width: int
    height: int

    def __init__(self, app: Application, file_vertex_shader: str, file_fragment_shader: str, width: int, height: int,
                 standalone=True, require=460) -> None:
        super().__init__()
        self.app = app
        self.file_fragment_shader = file_fragment_shader    # kept for signature compatibility; no GLSL here
        self.file_vertex_shader = file_vertex_shader
        self.width = width
        self.height = height
        self.standalone = standalone
        self.initialise()

    def initialise(self) -> None:
        raise NotImplementedError

    def configure_program(self, scene: Scene) -> None:
        raise NotImplementedError

    def configure_buffers_and_uniforms(self, scene: Scene) -> None:
        raise NotImplementedError

    def calculate_frame(self, scene: Scene) -> None:
        raise NotImplementedError

    def handle_event(self, event, app: Application, cam: Camera, scene: Scene, keyboard_enabled: bool, mouse_enabled: bool):
        raise NotImplementedError


class RayTracerDynamic(ProgramABC):
    """The path tracer program; every heavy step runs in librt_hip.so on one MI355X."""

    name = "raytracer"

    def __init__(self, app: Application, file_vertex_shader: str, file_fragment_shader: str, width: int, height: int,
                 standalone=not HEADLESS, require=460, lib: ffi.RtLib | None = None, device: int = 0,
                 save_frames: bool = True, export_root: str | Path = "renders/RayTracerDynamic", cull: bool = False,
                 sah: bool = False, display_png: bool = False) -> None:
        self._lib = lib                       # tests inject a checker library; the product loads the HIP one
        self._device = device
        self.save_frames = save_frames
        self.cull = cull                      # distance culling: same image, fewer box tests
        self.sah = sah                        # traverse the library's SAH tree over the same triangles: same image
        self.display_png = display_png        # also write the tone-mapped + gamma-corrected frame (`_tm.png`, to_display)
        self.export_root = Path(export_root)
        self.MAX_RAY_BOUNCES = 8
        self.RAYS_PER_PIXEL = 32
        self.is_scene_static = True
        self.ctx: ffi.RtContext | None = None
        self.last_image: np.ndarray | None = None
        super().__init__(app, file_vertex_shader, file_fragment_shader, width, height, standalone, require)

    def initialise(self) -> None:
        if self._lib is None:
            self._lib = ffi.RtLib()           # raises if the HIP extension has not been built
        if self.ctx is not None:
            self.ctx.close()
        self.ctx = self._lib.create(self._device)
        self.ctx.set_cull(self.cull)
        self.ctx.set_accel(self.sah)
        self._scene_uploaded = None

    # -- application.py:849-915 ---------------------------------------------------------------
    def configure_program(self, scene: Scene) -> None:
        self.ctx.resize(self.width, self.height)
        self.cycle_counter = 0
        self._scene_uploaded = None
        self.configure_buffers_and_uniforms(scene)

    # -- application.py:919-1002 --------------------------------------------------------------
    def configure_buffers_and_uniforms(self, scene: Scene) -> None:
        if self._scene_uploaded is not scene:
            # rest-pose geometry, topology and materials once; poses are applied on the device
            b = functions.export_scene_buffers(scene, rest_pose=True)
            self.ctx.set_scene(b["tris28"], b["meshes128"], b["nodes48"], b["leaf_tri_ids"], b["spheres96"], b["atmosphere80"])
            for mesh in scene.meshes:
                mesh.flag_for_mesh_update()
            self._scene_uploaded = scene
        for mesh in scene.meshes:
            if mesh.is_awaiting_mesh_update:   # numba transform + refit + re-pack, on the GPU
                self.ctx.update_mesh_pose(mesh.id, mesh.csys.pos, mesh.csys.quat)
                mesh.unflag_for_mesh_update()
        # animated materials (UBO 10/11 re-upload in the reference); node0Id/bbox fields are not read
        recs = np.zeros(len(Mesh.ALL), dtype=MESH_DTYPE)
        for i, m in enumerate(Mesh.ALL):
            recs[i]["index"] = m.id
            recs[i]["material"] = m.material.as_record()
        self.ctx.set_materials(recs, scene.atmosphere_material.as_record().reshape(1))
        self.ctx.set_max_cycles(self.app.MAX_CYCLES)      # program["MAX_CYCLES"] (application.py:997)
        self.app.is_animating = True
        if not hasattr(self, "app_frame_counter"):
            self.app_frame_counter = 0

    # -- application.py:1151-1223 -------------------------------------------------------------
    def calculate_frame(self, scene: Scene) -> None:
        cam = scene.cam
        self.app_frame_counter += 1
        self.app_frame_counter = SCENE.get_frame_number(scene, self.app_frame_counter)
        self.app.animate_frame(self.app_frame_counter)
        self.ctx.load_camera(cam)
        self.ctx.reset_accum()
        cam.cycle_counter = 0
        total = int(cam.passes_per_frame)
        while cam.cycle_counter < total:
            if self.app.is_requesting_exit.is_set():
                return
            # run up to the next cycle the reference would save (1..9, then every 10th); the running
            # mean continues across rt_render calls exactly (pass index = frameNumber)
            c = cam.cycle_counter
            if not self.save_frames:
                nxt = total
            elif c < 9:
                nxt = c + 1
            else:
                nxt = min(total, (c // 10 + 1) * 10)
            self.ctx.render(c + 1, nxt - c, 1, int(cam.rays_per_pixel), int(cam.bounces_per_ray), ffi.RT_ACCUM_MEAN)
            cam.cycle_counter = nxt
            if self.save_frames and (nxt < 10 or nxt % 10 == 0):
                self.save_frame(None, self.app_frame_counter, nxt)
        if not getattr(self, "defer_readback", False):    # a pipelined runner copies the frame back itself, asynchronously
            self.last_image = self.ctx.read_mean()
        if self.is_scene_static:
            cam.cycle_counter += 1
        else:
            self.is_scene_static = True
            cam.cycle_counter = 0

    # -- application.py:1094-1126 -------------------------------------------------------------
    def save_frame(self, fbo, frame, cycle) -> None:
        """PNG (clip -> *255 -> uint8, row 0 = top, application.py:1104-1117) + the float32 image as
        OpenEXR with the reference's file name (``_f4.exr``, :1117; written by ``write_exr``, imageio is
        absent here), and additionally as .npy and PFM (bottom-up scanlines, like fbo.read)."""
        from PIL import Image

        if not hasattr(self, "target_dir"):
            self.target_dir = self.export_root / datetime.datetime.now().strftime("%Y.%m.%d_%H%M%S")
            self.target_dir.mkdir(parents=True, exist_ok=True)
        hdr = self.ctx.read_mean()                       # row 0 = bottom scanline, like fbo.read
        flipped = np.flip(hdr, axis=0)
        rgb8 = (np.clip(flipped, 0.0, 1.0) * 255).astype(np.uint8)
        stem = f"f{frame:05}_c{cycle:05}"
        Image.fromarray(rgb8, "RGB").save(self.target_dir / f"{stem}_f1.png")
        np.save(self.target_dir / f"{stem}_f4.npy", flipped)
        write_pfm(self.target_dir / f"{stem}_f4.pfm", hdr)
        write_exr(self.target_dir / f"{stem}_f4.exr", flipped)
        if self.display_png:
            Image.fromarray(to_display(flipped), "RGB").save(self.target_dir / f"{stem}_tm.png")

    def handle_event(self, event, app, cam, scene, keyboard_enabled, mouse_enabled):
        return
