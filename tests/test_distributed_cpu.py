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


# ---- frame sharding: BASELINE.json configs[2] ------------------------------------------------------
FRAMES, AW, AH, APASSES, ABOUNCES = 5, 32, 18, 3, 4


def _anim_module():
    import types

    from raytracer_amd.scripts.animate import FrameAnimation
    from raytracer_amd.scripts.scenes import final_scene_numba as S

    sc = S.build()

    def override(cam, i):
        cam.passes_per_frame, cam.bounces_per_ray = APASSES, ABOUNCES

    sc.animations.append(FrameAnimation(sc.cam, override))
    return types.SimpleNamespace(scene=sc, get_frame_number=S.get_frame_number, __file__=S.__file__)


def _anim_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    from conftest import ORACLE_SO
    from raytracer_amd.distributed import ShardedAnimation
    from raytracer_amd.ffi import RtLib

    dist.init_process_group("gloo", rank=rank, world_size=world)
    anim = ShardedAnimation(_anim_module(), AW, AH, lib=RtLib(ORACLE_SO))
    got = anim.run(FRAMES, gather=True, first_frame=40)
    np.savez(Path(out_dir) / f"frames_{rank}.npz", **{str(f): img for f, img in got.items()})
    anim.close()
    dist.destroy_process_group()


def test_frame_sharding_world2_gloo(tmp_path, oracle_lib):
    """rank r animates and renders frames 40+r, 40+r+2, ... through Application.animate_frame /
    calculate_frame; the union (gathered to rank 0) equals the single-process frames bit for bit."""
    import torch.multiprocessing as mp

    from raytracer_amd.distributed import ShardedAnimation

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_anim_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "frames_0.npz"), np.load(tmp_path / "frames_1.npz")
    assert sorted(int(k) for k in r0.files) == [40, 41, 42, 43, 44]          # rank 0 holds every frame after the gather
    assert sorted(int(k) for k in r1.files) == [41, 43]                      # rank 1 rendered (and keeps) the odd ones
    single = ShardedAnimation(_anim_module(), AW, AH, lib=oracle_lib)
    ref = single.run(FRAMES, first_frame=40)
    single.close()
    assert sorted(ref) == [40, 41, 42, 43, 44]
    for f in ref:
        assert ref[f].max() > 0
        assert np.array_equal(ref[f].view(np.uint32), r0[str(f)].view(np.uint32)), f"frame {f} differs"
    for f in (41, 43):
        assert np.array_equal(r1[str(f)].view(np.uint32), r0[str(f)].view(np.uint32))
    assert not np.array_equal(ref[40], ref[44])                               # the animation moves
    # and a frame rendered out of order equals the same frame rendered in sequence (frame purity, application.py:1167-1172)
    from conftest import build_suzanne, make_ctx

    c = make_ctx(oracle_lib, build_suzanne(43), AW, AH)
    c.render(1, APASSES, 1, 1, ABOUNCES, 0)
    assert np.array_equal(c.read_mean().view(np.uint32), ref[43].view(np.uint32))


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_launches_itself_for_n_gpus(launcher):
    """`python bench.py --gpus 2` with no launcher starts its two ranks itself (torch.distributed.run children of a parent
    that never touches a GPU); the driver's shape - torch.distributed.run around bench.py - keeps working.  CPU-only: the
    ranks meet over gloo, run one collective and print rank 0's line (RT_BENCH_LAUNCH_CHECK=1) before the GPU is needed;
    the full two-rank step runs on the GPU box (tests/test_gpu_rccl_single_rank.py)."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_BENCH_LAUNCH_CHECK"] = "1"
    args = [str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", "strong"]
    if launcher == "self":
        cmd = [sys.executable, *args]
    else:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), *args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["launch_check"] and d["n_gpus"] == 2 and d["rank_sum"] == 3 and d["steps"] == 3 and d["scaling"] == "strong"


def test_bench_self_launch_forwards_a_failing_exit_code():
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_BENCH_LAUNCH_CHECK"] = "1"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--workload", "no-such-workload"], env=env,
                       capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode != 0
