"""BASELINE.json configs[2] in the driver-run suite: the keyframe animation of final_scene_numba
(input/camera_keyframes.csv, 250 rows), every 25th frame plus the frames where the scene's material hooks
(reference: scripts/scenes/final_scene_numba.py:228-353) switch something on or off:

  20, 50   rear wall: blue diffuse -> mirror ramp starts / ends (:228-247)
  49..51   front wall: alpha ramp 0 -> 1 starts at 50 (:265-283): frame 49 is the last with alpha 0 (every ray
           passes through, :725-736), 51 the first with a draw against alpha; light emission ramp starts (:348-352)
  65       mid-ramp: alpha pass-through draws on the front wall, transparentFromBehind on, light ramping
  79..81   alpha reaches 1 (no more pass-through draw), the blue wall starts turning into a mirror (:287-306)
  95       front wall fully specular; white_upper ramp under way (:311-317)
  119      last keyframe of the 120-row dragon CSV range; red ramp about to start

One context per mode lives across ALL frames and is brought up to date per frame the way the application
does it (pose of the flagged meshes, every material, camera: application.py:939-976, 1175-1183) - so the device
pose/refit, the materials-only path (k_tri_flags without a refit) and the running state are what is tested,
not a fresh upload per frame.  HIP exact mode and HIP SAH+culled (the bench mode) against the oracle,
320x180, 2 passes x 1 ray, 8 bounces (the scene's own bounce count, :331): bit for bit."""
import numpy as np
import pytest

from conftest import build_suzanne

pytestmark = pytest.mark.gpu

W, H, PASSES, BOUNCES = 320, 180, 2, 8
FRAMES = sorted(set(range(0, 250, 25)) | {20, 49, 50, 51, 65, 79, 80, 81, 95, 119, 249})


def _ctx(lib, sc, sah, cull):
    c = lib.create(0)
    if lib.backend == "hip":
        c.set_accel(sah)
    c.load_scene(sc)
    c.resize(W, H)
    if lib.backend == "hip":
        c.set_cull(cull)
    return c


def _frame(c, sc):
    c.update_scene(sc)
    c.load_camera(sc.cam)
    ids, ts = c.first_hit(1)
    c.reset_accum()
    c.render(1, PASSES, 1, 1, BOUNCES, 0)
    return ids, ts, c.read_mean()


def test_keyframes_exact_and_bench_mode_against_the_oracle(oracle_lib, hip_lib):
    scenes = [build_suzanne(0) for _ in range(3)]
    ctxs = [_ctx(oracle_lib, scenes[0], False, False), _ctx(hip_lib, scenes[1], False, False), _ctx(hip_lib, scenes[2], True, True)]
    alpha_seen, moved = set(), 0
    prev_ids = None
    for f in FRAMES:
        out = []
        for sc, c in zip(scenes, ctxs):
            sc.animate_frame(f)
            out.append(_frame(c, sc))
        (io, to, ao), (ig, tg, ag), (isah, tsah, asah) = out
        front = scenes[0].materials["front"]
        alpha_seen.add(round(float(front.color[3]), 3))
        assert (io >= 0).mean() > 0.3, f
        for name, (i2, t2, a2) in (("exact", (ig, tg, ag)), ("SAH+culled", (isah, tsah, asah))):
            assert np.array_equal(io, i2), f"frame {f}, {name}: {(io != i2).sum()} first-hit ids differ"
            assert np.array_equal(to.view(np.uint32), t2.view(np.uint32)), f"frame {f}, {name}: first-hit distances differ"
            assert np.array_equal(ao.view(np.uint32), a2.view(np.uint32)), f"frame {f}, {name}: {(ao != a2).any(axis=2).sum()} pixels differ"
        if prev_ids is not None and not np.array_equal(prev_ids, io):
            moved += 1
        prev_ids = io
    assert moved >= len(FRAMES) - 2                      # the camera and the monkey really move between the frames
    assert 0.0 in alpha_seen and 1.0 in alpha_seen and any(0.0 < a < 1.0 for a in alpha_seen)   # all three alpha regimes were rendered
    for c in ctxs:
        c.close()


_PIPE_CHILD = r"""
import sys, types
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import torch
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")                      # torch brings the HIP runtime up first (INTEGRATION.md 4)
from raytracer_amd.distributed import ShardedAnimation
from raytracer_amd.ffi import RtLib
from raytracer_amd.scripts.animate import FrameAnimation
from raytracer_amd.scripts.scenes import final_scene_numba as S

def module():
    sc = S.build()
    def override(cam, i):
        cam.passes_per_frame, cam.bounces_per_ray = 3, 4
    sc.animations.append(FrameAnimation(sc.cam, override))
    return types.SimpleNamespace(scene=sc, get_frame_number=S.get_frame_number, __file__=S.__file__)

lib = RtLib()
seq = ShardedAnimation(module(), 96, 54, lib=lib, cull=True, sah=True)
want = seq.run(5, first_frame=118)
seq.close()
pipe = ShardedAnimation(module(), 96, 54, lib=lib, cull=True, sah=True)
seen = []
got = pipe.run_pipelined(5, first_frame=118, on_frame=lambda f, img: seen.append(f))
assert sorted(got) == sorted(want) == [118, 119, 120, 121, 122] and seen == [118, 119, 120, 121, 122], (sorted(got), seen)
for f in want:
    assert want[f].max() > 0 and np.array_equal(want[f].view(np.uint32), got[f].view(np.uint32)), f
assert not np.array_equal(want[118], want[122])
again = pipe.render_frame(120)                     # the context's own accumulator and stream again
assert np.array_equal(again.view(np.uint32), want[120].view(np.uint32))
pipe.close()
print("PIPELINE_OK")
"""


@pytest.mark.gpu
def test_pipelined_animation_equals_the_sequential_one():
    """ShardedAnimation.run_pipelined (two accumulators bound in turn, asynchronous read-back behind the render on torch's
    stream) returns the frames of ShardedAnimation.run bit for bit, and leaves the context usable afterwards.  In a child
    interpreter that initialises torch first, like tests/test_gpu_torch_binding.py."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, "-c", _PIPE_CHILD, str(root)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PIPELINE_OK" in r.stdout, r.stderr[-3000:]
