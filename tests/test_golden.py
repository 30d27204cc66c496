"""Committed golden vectors (tests/golden/, made by make_golden.py from the oracle)."""
from pathlib import Path

import numpy as np
import pytest

from conftest import build_suzanne, make_ctx

G = Path(__file__).resolve().parent / "golden"


def _check(lib, cull=False):
    g = np.load(G / "suzanne_f0_64x36_firsthit_pass1.npz")
    c = make_ctx(lib, build_suzanne(0), 64, 36)
    c.set_cull(cull)
    ids, ts = c.first_hit(1)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(ts.view(np.uint32), g["t"].view(np.uint32))
    g = np.load(G / "suzanne_f0_32x18_mean_p1-3_b4.npz")
    c = make_ctx(lib, build_suzanne(0), 32, 18)
    c.set_cull(cull)
    assert np.array_equal(c.read_tris28().view(np.uint32), g["tris28"].view(np.uint32))
    c.render(1, 3, 1, 1, 4, 0)
    assert np.array_equal(c.read_mean().view(np.uint32), g["mean"].view(np.uint32))
    g = np.load(G / "suzanne_f100_32x18_mean_p1-2_rpp2_b8.npz")
    c = make_ctx(lib, build_suzanne(100), 32, 18)
    c.set_cull(cull)
    c.render(1, 2, 1, 2, 8, 0)
    assert np.array_equal(c.read_mean().view(np.uint32), g["mean"].view(np.uint32))


def test_oracle_reproduces_golden(oracle_lib):
    _check(oracle_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("cull", [False, True])
def test_hip_reproduces_golden(hip_lib, cull):
    _check(hip_lib, cull)
