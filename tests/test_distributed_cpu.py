"""N > 1 path on CPU: world_size-2 gloo, the oracle standing in for the GPU library behind
the same ShardedRenderer code (pass sharding, SUM accumulation, one all-reduce per frame)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_shard_passes_partition():
    from raytracer_amd.distributed import frames_of_rank, shard_passes

    for total in (1, 5, 64, 67):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, n, stride = shard_passes(total, r, world)
                seen += [first + i * stride for i in range(n)]
            assert sorted(seen) == list(range(1, total + 1))
    assert sorted(sum((list(frames_of_rank(250, r, 8)) for r in range(8)), [])) == list(range(250))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    from conftest import ORACLE_SO, build_suzanne
    from raytracer_amd.distributed import ShardedRenderer
    from raytracer_amd.ffi import RtLib

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = build_suzanne(0)
    ctx = RtLib(ORACLE_SO).create()
    ctx.load_scene(sc)
    ctx.load_camera(sc.cam)
    r = ShardedRenderer(ctx, 48, 27)
    mean = r.render(6, 1, 4).numpy().copy()
    np.save(Path(out_dir) / f"mean_{rank}.npy", mean)
    dist.destroy_process_group()


def test_spp_sharding_world2_gloo(tmp_path, oracle_lib):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = np.load(tmp_path / "mean_0.npy"), np.load(tmp_path / "mean_1.npy")
    assert np.array_equal(m0, m1)                       # every rank holds the same image
    # single process, same 6 passes: SUM in pass order, and the reference's running mean
    from conftest import build_suzanne, make_ctx

    c = make_ctx(oracle_lib, build_suzanne(0), 48, 27)
    c.render(1, 6, 1, 1, 4, 1)
    single_sum_mean = c.read_mean()
    c = make_ctx(oracle_lib, build_suzanne(0), 48, 27)
    c.render(1, 6, 1, 1, 4, 0)
    running_mean = c.read_mean()
    assert m0.max() > 0
    # only float summation order differs: (p1+p3+p5)+(p2+p4+p6) vs ((p1+p2)+...) vs running mean
    tol = 1e-5 * max(1.0, float(running_mean.max()))
    assert np.abs(m0 - single_sum_mean).max() <= tol
    assert np.abs(m0 - running_mean).max() <= tol
