"""Shared fixtures.  GPU tests are marked ``gpu`` and call the product through its C-ABI
(raytracer_amd/librt_hip.so); everything else runs on the CPU.  The oracle
(oracle/librt_oracle.so) is the checker, never the thing under test in a ``gpu`` test."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

ORACLE_SO = ROOT / "oracle" / "librt_oracle.so"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from raytracer_amd.ffi import RtLib

    if not ORACLE_SO.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle")])
    return RtLib(ORACLE_SO)


@pytest.fixture(scope="session")
def hip_lib():
    from raytracer_amd.ffi import RtLib

    return RtLib()  # raises loudly if librt_hip.so is missing


@pytest.fixture(scope="session")
def hip_lib_cpu():
    """The product library for its host-only entry points (rt_bvh_*): loads without a GPU."""
    from raytracer_amd.ffi import RtLib

    return RtLib()


def build_suzanne(frame=0):
    from raytracer_amd.scripts.scenes import final_scene_numba as S

    sc = S.build()
    sc.animate_frame(frame)
    return sc


@pytest.fixture()
def suzanne():
    return build_suzanne(0)


def make_ctx(lib, scene, w, h, rest_pose=True):
    ctx = lib.create()
    ctx.load_scene(scene, rest_pose=rest_pose)
    ctx.load_camera(scene.cam)
    ctx.resize(w, h)
    return ctx
