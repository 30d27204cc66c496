"""The RCCL leg of the N > 1 paths on the one GPU a test box has: a process group of ONE rank on the `nccl` backend
(= RCCL).  Two ranks cannot share a card under RCCL, so this is as far as a one-GPU box can go: it runs
`init_process_group("nccl", device_id=...)`, the all-reduce of the bound accumulator on the render's stream, the barriers
and `set_accum_passes` - everything `bench.py --gpus N` and ShardedRenderer / ShardedAnimation do per rank - and checks
the image against the single-process render.  (World size 2 is covered on CPU with gloo: test_distributed_cpu.py.)"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    return env


@pytest.mark.gpu
def test_bench_takes_the_rccl_path_with_one_rank():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=_env(RT_BENCH_FORCE_DIST="1", RT_BENCH_REHEARSAL="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "RCCL all-reduce" in line["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_self_launch_two_ranks_share_the_card():
    """`python bench.py --gpus 2` with no launcher, on the one GPU of the box: the parent starts both ranks itself; they share
    the card (gloo between them, tiny frames, turns: RT_BENCH_REHEARSAL), walk the whole N = 2 step - sharded passes, SUM
    accumulation, all-reduce, max-over-ranks timing - and rank 0's ONE line comes back with n_gpus == 2.  Not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RT_BENCH_REHEARSAL="1", RT_BENCH_BACKEND="gloo", RT_BENCH_WATCHDOG="200", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for scaling in ("weak", "strong"):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", scaling],
                           env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["value"] > 0 and "rehearsal" in line
        assert "spp-sharded x2" in line["config"]["parallelism"]


_SHARDED = r"""
import sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from conftest import build_suzanne
from raytracer_amd.distributed import ShardedRenderer
from raytracer_amd.ffi import RtLib
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
lib = RtLib()
sc = build_suzanne(0)
ctx = lib.create(0)
ctx.load_scene(sc); ctx.load_camera(sc.cam)
r = ShardedRenderer(ctx, 96, 54)
mean = r.render(6, 1, 4)
torch.cuda.synchronize()
assert ctx.read_accum()[1] == 6
np.savez({out!r}, total=r.accum.cpu().numpy(), mean=mean.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_sharded_renderer_on_rccl_equals_the_sum_mode_render(tmp_path, hip_lib):
    from conftest import build_suzanne, make_ctx

    out = tmp_path / "img.npz"
    code = _SHARDED.format(root=str(ROOT), tests=str(ROOT / "tests"), out=str(out))
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    c = make_ctx(hip_lib, build_suzanne(0), 96, 54)
    c.render(1, 6, 1, 1, 4, 1)                       # RT_ACCUM_SUM over the same six passes, one rank's shard = all of them
    want, n = c.read_accum()
    assert n == 6 and got["total"].max() > 0
    assert np.array_equal(got["total"].view(np.uint32), want.view(np.uint32))          # the all-reduce of one rank changes nothing
    assert np.abs(got["mean"] - c.read_mean()).max() <= 1e-6 * max(1.0, float(want.max()))   # torch divides by multiplying with 1/n


@pytest.mark.gpu
def test_animation_runner_gathers_over_rccl_with_one_rank():
    r = subprocess.run([sys.executable, "-m", "raytracer_amd.distributed", "--frames", "3", "--first-frame", "60", "--height", "90",
                        "--passes", "2", "--gather"], cwd=str(ROOT), env=_env(RT_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["frames"] == 3 and line["frames_held_by_rank0"] == 3 and line["value"] > 0


_C_COLLECTIVE = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from conftest import build_suzanne
from raytracer_amd.ffi import RT_ACCUM_SUM, RtError, RtLib

lib = RtLib()
sc = build_suzanne(0)
ctx = lib.create(0)                        # hipSetDevice(0) before the communicator is made
ctx.load_scene(sc); ctx.load_camera(sc.cam); ctx.resize(96, 54)
rccl = ctypes.CDLL("librccl.so.1", mode=ctypes.RTLD_GLOBAL)     # what a C host links; rt_allreduce_sum finds this instance
class Uid(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]
uid, comm = Uid(), ctypes.c_void_p()
assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
ctx.render(1, 3, 2, 1, 4, RT_ACCUM_SUM)    # the shard of rank 0 of 2: passes 1, 3, 5
before = ctx.read_accum()[0].copy()
ctx.allreduce_sum(comm.value, 3)           # on the context's stream, no host synchronisation in between
after, n = ctx.read_accum()
assert n == 3 and before.max() > 0 and np.array_equal(before.view(np.uint32), after.view(np.uint32))
assert np.array_equal(ctx.read_mean(), before / np.float32(3))
ctx.reset_accum(); ctx.render(1, 2, 1, 1, 4, 0)
try:
    ctx.allreduce_sum(comm.value, 2)       # a running mean cannot be summed over ranks
    raise SystemExit("a mean-mode accumulator was accepted")
except RtError as e:
    assert e.code == -2, e
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclCommDestroy(comm)
ctx.close()
print("c-collective ok")
"""


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_c_abi_collective_over_an_rccl_communicator_of_one_rank():
    """rt_allreduce_sum: the exchange step for a host without torch - an ncclComm_t made through RCCL's own C
    API, the all-reduce issued by the library on the render's stream.  One rank is what a one-GPU box can do: the sum over one
    rank leaves the accumulator bit for bit, the pass count becomes the job's."""
    code = _C_COLLECTIVE.format(root=str(ROOT), tests=str(ROOT / "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "c-collective ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
