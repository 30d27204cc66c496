"""BASELINE.json configs[1] at its FULL size - Suzanne room, 1920x1080, 64 passes x 1 ray per pixel, 4 bounces,
6.4e8 closest-hit queries - rendered by the CPU oracle (about 27 s on 16 threads) and by the HIP library in its
exact mode and in the mode bench.py measures (SAH structure + culling): first-hit ids, distances and the float
bit patterns of the mean image are identical, and so are the ray / hit / sample counters (and the triangle-test
counter of the exact mode).  tests/full_size_oracle.py is the same check as a script."""
import numpy as np
import pytest

from conftest import build_suzanne

W, H, PASSES, BOUNCES = 1920, 1080, 64, 4


def _render(lib, sah, cull):
    sc = build_suzanne(0)
    c = lib.create()
    c.set_accel(sah)
    c.load_scene(sc)
    c.load_camera(sc.cam)
    c.resize(W, H)
    c.set_cull(cull)
    c.enable_stats(True)
    ids, ts = c.first_hit(1)
    c.render(1, PASSES, 1, 1, BOUNCES, 0)
    out = ids, ts, c.read_mean(), c.stats()
    c.close()
    return out


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_configuration_full_size_against_the_oracle(oracle_lib, hip_lib):
    ref = _render(oracle_lib, False, False)
    assert ref[3]["rays"] == 634615464 + W * H          # the 64 passes + the first-hit probe
    for name, sah, cull in (("exact", False, False), ("SAH culled", True, True)):
        r = _render(hip_lib, sah, cull)
        assert np.array_equal(ref[0], r[0]), f"{name}: first-hit ids"
        assert np.array_equal(ref[1].view(np.uint32), r[1].view(np.uint32)), f"{name}: first-hit distances"
        assert np.array_equal(ref[2].view(np.uint32), r[2].view(np.uint32)), f"{name}: {(ref[2] != r[2]).any(axis=2).sum()} pixels differ"
        for k in ("rays", "shaded_hits", "samples"):
            assert ref[3][k] == r[3][k], (name, k)
        if name == "exact":
            assert ref[3]["tri_tests"] == r[3]["tri_tests"]
