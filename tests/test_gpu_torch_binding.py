"""The accumulator bound to a torch tensor, kernels on torch's stream (what bench.py and
ShardedRenderer do before the RCCL all-reduce): a torch op enqueued right after rt_render,
with no host synchronisation in between, must see the finished image - on torch's default
stream (HIP's null stream, handle 0) and on a side stream.

Runs in a child interpreter that initialises torch first, like bench.py: torch ships its own
copy of the HIP runtime and must be the one that brings it up."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import torch
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
from raytracer_amd.ffi import RtLib
from conftest import build_suzanne, make_ctx
W, H = 480, 270
lib = RtLib()
ref = make_ctx(lib, build_suzanne(), W, H)
ref.set_cull(True)
ref.render(1, 8, 1, 1, 4, 0)
want = ref.read_mean()
for side_stream in (False, True):
    ctx = make_ctx(lib, build_suzanne(), W, H)
    ctx.set_cull(True)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.Stream(dev) if side_stream else torch.cuda.current_stream(dev)
    assert side_stream or stream.cuda_stream == 0, "torch's default stream is expected to be the null stream"
    own = ctx.stream()
    with torch.cuda.stream(stream):
        ctx.set_stream(stream.cuda_stream)
        ctx.bind_accum(accum.data_ptr())
        assert ctx.stream() == stream.cuda_stream
        for _ in range(3):                       # repeated: a race would show as a partial image
            ctx.reset_accum()
            ctx.render(1, 8, 1, 1, 4, 0)
            snap = accum.clone()                 # enqueued behind the render, no host sync
            got = snap.cpu().numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"side_stream={side_stream}: image not complete"
    ctx.set_stream(None)                         # back to the context's own stream
    assert ctx.stream() == own
    ctx.bind_accum(None)
    ctx.close()
print("OK")
'''


@pytest.mark.gpu
def test_render_is_ordered_on_torchs_stream(hip_lib):
    r = subprocess.run([sys.executable, "-c", CHILD, str(ROOT)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
