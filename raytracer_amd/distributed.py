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
        if self.world > 1:
            self.dist.all_reduce(self.accum, group=self.group)   # RCCL over xGMI on GPU, gloo in tests
        self.ctx.set_accum_passes(total_passes)
        return self.accum / float(total_passes)

    def mean_numpy(self, total_passes: int) -> np.ndarray:
        return (self.accum / float(total_passes)).cpu().numpy()
