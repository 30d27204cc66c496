"""Multi-GPU sharding of the render path: one process per GPU, torch.distributed (RCCL) only
for the one exchange the path really has.

The reference is single-GPU; what makes the path shard is in its shader: every
(pixel, pass) sample is seeded independently (shaders/raytracer.fs.glsl:956-957) and a frame
is a pure function of its index (scripts/application.py:1167-1172).  Hence

* spp sharding  - rank r of n renders pass indices r+1, r+1+n, ... of a frame into a SUM
  accumulator (rt_render with pass_stride = n); ONE all-reduce of the W*H*3 float32 sum per
  frame combines them (24.9 MB at 1080p, 99.5 MB at 4K; ring all-reduce over xGMI moves
  2(n-1)/n of that per GPU) and every rank holds the mean = sum / passes;
* frame sharding - frame f goes to rank f mod n; no collective at all.

The accumulator lives in a torch tensor (device memory + NCCL plumbing); the kernels write
into it through rt_bind_accum and run on torch's current stream (rt_set_stream), so the
all-reduce is ordered after the render without a host synchronisation.
"""
from __future__ import annotations

import numpy as np

from . import ffi


def shard_passes(total_passes: int, rank: int, world: int) -> tuple[int, int, int]:
    """(first_pass, n_passes, stride) of `rank`: passes rank+1, rank+1+world, ... <= total."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    n = (total_passes - rank + world - 1) // world if total_passes > rank else 0
    return rank + 1, n, world


def frames_of_rank(n_frames: int, rank: int, world: int) -> range:
    """Frame indices rendered by `rank` (round-robin)."""
    return range(rank, n_frames, world)


class ShardedRenderer:
    """spp-sharded rendering of one frame across the ranks of a process group."""

    def __init__(self, ctx: ffi.RtContext, width: int, height: int, group=None, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.ctx, self.group = ctx, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.on_gpu = ctx.lib.backend == "hip"
        if self.on_gpu:
            device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            self.accum = torch.zeros((height, width, 3), dtype=torch.float32, device=device)
            ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
        else:  # CPU oracle under gloo (tests): host memory shared between numpy, torch and the C library
            self.accum = torch.zeros((height, width, 3), dtype=torch.float32)
        ctx.resize(width, height)
        ctx.bind_accum(self.accum.data_ptr())

    def render(self, total_passes: int, rays_per_pixel: int, max_bounces: int):
        """Render passes 1..total_passes of the current scene/camera, sharded; returns the
        mean image tensor (identical on every rank)."""
        first, n, stride = shard_passes(total_passes, self.rank, self.world)
        self.accum.zero_()
        self.ctx.bind_accum(self.accum.data_ptr())     # also resets the context's pass count
        if n:
            self.ctx.render(first, n, stride, rays_per_pixel, max_bounces, ffi.RT_ACCUM_SUM)
        if not self.on_gpu:
            self.ctx.sync()
        if self.dist.is_initialized():                           # also with one rank: the same code path as with eight
            self.dist.all_reduce(self.accum, group=self.group)   # RCCL over xGMI on GPU, gloo in tests
        self.ctx.set_accum_passes(total_passes)
        return self.accum / float(total_passes)

    def mean_numpy(self, total_passes: int) -> np.ndarray:
        return (self.accum / float(total_passes)).cpu().numpy()


class ShardedAnimation:
    """Frame-sharded rendering of a keyframe animation (BASELINE.json configs[2]): rank r of n
    animates and renders frames r, r+n, ... through the reference's own call path -
    ``Application.animate_frame`` -> ``RayTracerDynamic.calculate_frame`` (application.py:390-394,
    1151-1223) - on its own GPU.  A frame is a pure function of its index (``get_frame_number`` +
    ``Scene.animate_frame``, application.py:1167-1172), so there is no data-path collective; finished
    frames are written by the rank that rendered them and can be gathered to rank 0, one round of n
    frames at a time (bounded memory).

    ``scene_module`` must expose ``scene`` and ``get_frame_number`` like the reference's scene modules.
    """

    def __init__(self, scene_module, width: int, height: int, lib: ffi.RtLib | None = None, device: int = 0, group=None,
                 cull: bool = False, sah: bool = False, save_frames: bool = False, export_root="renders/RayTracerDynamic"):
        from .scripts import application as A

        self.A = A
        self.group = group
        try:
            import torch.distributed as dist

            self.dist = dist if dist.is_initialized() else None
        except ImportError:                      # single process without torch
            self.dist = None
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        A.set_scene(scene_module)
        self.app = A.Application.__new__(A.Application)
        self.prog = A.RayTracerDynamic.__new__(A.RayTracerDynamic)
        A.RayTracerDynamic.__init__(self.prog, self.app, file_vertex_shader="shaders/raytracer.vs.glsl",
                                    file_fragment_shader="shaders/raytracer.fs.glsl", width=width, height=height, lib=lib,
                                    device=device, save_frames=save_frames, export_root=export_root, cull=cull, sah=sah)
        A.Application.__init__(self.app, render_height=height, max_frames=None, program=self.prog)
        self.prog.configure_program(self.app.display_scene)
        self.width, self.height = width, height

    def render_frame(self, frame: int) -> np.ndarray:
        """One animation frame through the program's own per-frame entry (the counter is pre-set so that
        ``app_frame_counter += 1`` lands on `frame`, application.py:1167-1169)."""
        self.prog.app_frame_counter = frame - 1
        self.prog.calculate_frame(self.app.display_scene)
        return self.prog.last_image

    def run_pipelined(self, n_frames: int, first_frame: int = 0, on_frame=None) -> dict:
        """This rank's frames like ``run`` (no gather), with the read-back of frame k overlapped with the host work and
        the render of frame k+1: two device accumulators bound in turn (``rt_bind_accum``), the render on torch's
        stream, an asynchronous copy into pinned host memory behind it, and the host only waits for a frame when the
        next one is already queued.  Same images bit for bit (the accumulator of RT_ACCUM_MEAN is the mean).  HIP
        backend only: the checker library renders on the host, where there is nothing to overlap.  The process must have
        let torch initialise the GPU before the library was loaded, as the runner below does."""
        import torch

        prog, ctx = self.prog, self.prog.ctx
        if ctx.lib.backend != "hip" or prog.save_frames:
            return self.run(n_frames, first_frame=first_frame, on_frame=on_frame)
        dev = torch.device("cuda", torch.cuda.current_device())
        acc = [torch.zeros((self.height, self.width, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        host = [torch.empty((self.height, self.width, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
        done = [None, None]
        frame_in = [None, None]
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        out = {}

        def collect(b):
            if frame_in[b] is None:
                return
            done[b].synchronize()
            img = host[b].numpy().copy()
            out[frame_in[b]] = img
            if on_frame is not None:
                on_frame(frame_in[b], img)
            frame_in[b] = None

        prog.defer_readback = True
        try:
            mine = [first_frame + f for f in frames_of_rank(n_frames, self.rank, self.world)]
            for k, frame in enumerate(mine):
                b = k & 1
                collect(b)                                   # the frame that used this pair of buffers two frames ago
                ctx.bind_accum(acc[b].data_ptr())
                prog.app_frame_counter = frame - 1
                prog.calculate_frame(self.app.display_scene)   # animate, upload, render: all queued, nothing read back
                host[b].copy_(acc[b], non_blocking=True)
                done[b] = torch.cuda.Event()
                done[b].record()
                frame_in[b] = frame
                collect(1 - b)                               # the previous frame finishes while this one renders
            collect(0)
            collect(1)
        finally:
            prog.defer_readback = False
            torch.cuda.synchronize(dev)
            ctx.bind_accum(None)                             # back to the context's own accumulator and stream
            ctx.set_stream(None)
        prog.last_image = out[mine[-1]] if mine else None
        return out

    def run(self, n_frames: int, gather: bool = False, first_frame: int = 0, on_frame=None) -> dict:
        """Render frames first_frame .. first_frame+n_frames-1, sharded round-robin.  Returns {frame:
        image} of this rank's frames; with ``gather`` rank 0 returns every frame (other ranks their own).
        ``on_frame(frame, image)`` is called on the rank that holds the image (rank 0 when gathering)."""
        mine = [first_frame + f for f in frames_of_rank(n_frames, self.rank, self.world)]
        out = {}
        rounds = (n_frames + self.world - 1) // self.world
        grouped = self.dist is not None and self.dist.is_initialized()   # also a group of one rank: same code path as eight
        for k in range(rounds):
            img = None
            if k < len(mine):
                img = self.render_frame(mine[k])
                out[mine[k]] = img
                if on_frame is not None and not (gather and grouped):
                    on_frame(mine[k], img)
            if gather and grouped:
                got = self._gather_round(img)
                if self.rank == 0:
                    for r, g in enumerate(got):
                        f = first_frame + k * self.world + r
                        if f < first_frame + n_frames:
                            out[f] = g
                            if on_frame is not None:
                                on_frame(f, g)
        return out

    def _gather_round(self, img):
        import torch

        on_gpu = self.dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        t = torch.zeros((self.height, self.width, 3), dtype=torch.float32, device=dev)
        if img is not None:
            t.copy_(torch.from_numpy(np.ascontiguousarray(img)))
        bufs = [torch.empty_like(t) for _ in range(self.world)] if self.rank == 0 else None
        # `dst` is a GLOBAL rank while self.rank is the rank inside `group`: they differ for a sub-group without global rank 0
        dst = 0 if self.group is None else self.dist.get_global_rank(self.group, 0)
        self.dist.gather(t, bufs, dst=dst, group=self.group)
        return [b.cpu().numpy() for b in bufs] if self.rank == 0 else None

    def close(self) -> None:
        if self.prog.ctx is not None:
            self.prog.ctx.close()
            self.prog.ctx = None


def main(argv=None) -> None:
    """configs[2] runner: ``python -m torch.distributed.run --nproc-per-node N -m raytracer_amd.distributed
    --frames 250 --height 1080 --passes 64`` renders the keyframe animation of final_scene_numba, frames sharded
    over the ranks (one GPU each), and prints one JSON line with the whole-job frame rate on rank 0."""
    import argparse
    import json
    import os
    import time

    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=250)
    ap.add_argument("--first-frame", type=int, default=0)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--passes", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--save", action="store_true", help="write PNG/EXR/.npy of every frame at the reference's cadence")
    ap.add_argument("--gather", action="store_true", help="collect the finished frames on rank 0")
    ap.add_argument("--pipeline", action="store_true", help="overlap the read-back of a frame with the next frame (run_pipelined; no gather)")
    ap.add_argument("--reference-structure", action="store_true")
    a = ap.parse_args(argv)

    import types

    import torch
    import torch.distributed as dist

    from .scripts.animate import FrameAnimation
    from .scripts.scenes import final_scene_numba as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path with several ranks sharing one GPU (nccl = RCCL needs a GPU per rank)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    grouped = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST", "0") == "1"   # FORCE: the RCCL leg with one rank (tests)
    if grouped:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    sc = S.build()

    def override(cam, i):
        cam.passes_per_frame, cam.bounces_per_ray = a.passes, a.bounces

    sc.animations.append(FrameAnimation(sc.cam, override))
    mod = types.SimpleNamespace(scene=sc, get_frame_number=S.get_frame_number, __file__=S.__file__)
    anim = ShardedAnimation(mod, a.height * 16 // 9, a.height, device=local, cull=True, sah=not a.reference_structure,
                            save_frames=a.save)
    anim.render_frame(a.first_frame)                 # warm-up: first launch, allocations
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if a.pipeline and not a.gather:
        got = anim.run_pipelined(a.frames, first_frame=a.first_frame)
    else:
        got = anim.run(a.frames, gather=a.gather, first_frame=a.first_frame)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    dt = time.perf_counter() - t0
    if anim.rank == 0:
        print(json.dumps({"metric": "animation frames/s (configs[2]: keyframes of final_scene_numba, frames sharded over the ranks)",
                          "value": round(a.frames / dt, 3), "unit": "frames/s", "n_gpus": world, "frames": a.frames,
                          "frames_held_by_rank0": len(got), "seconds": round(dt, 3),
                          "config": {"workload": f"{a.height * 16 // 9}x{a.height}, {a.passes} passes x 1 ray/px, {a.bounces} bounces per frame, "
                                                 "host animate hooks + device pose/refit + render + read-back"}}), flush=True)
    anim.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
