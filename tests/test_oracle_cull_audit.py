"""The oracle's audit of the product's two distance-culling rules (oracle/rt_oracle.c: rto_cull_audit; rules in
raytracer_amd/csrc/rt_dev_trace.h: culled()).  For every closest-hit query it asks whether the box of the WINNER's leaf
(in the reference's tree) or the winner's own bounding box (inside any box of any structure over the same triangles,
hence the SAH structure's too) is one the rules could skip - independent of the order a traversal visits boxes in.  Zero flagged queries means a
culling traversal must find the reference's winner; a flagged query is one where float rounding put an accepted hit
(t >= 1e-6, hit point rounded onto a box face) outside the interval its own leaf box yields.  The reference's scenes
flag none; the grid-aligned scene with the camera 1e-6 beside the face planes does (the limit of
'culled == exact')."""
import ctypes

import numpy as np

from conftest import ORACLE_SO, build_suzanne, make_ctx
from test_gpu_fuzz import _grid_scene, _render


def _audit(mode):
    lib = ctypes.CDLL(str(ORACLE_SO))
    out = (ctypes.c_ulonglong * 5)()
    lib.rto_cull_audit(ctypes.c_int(mode), out)
    return [int(v) for v in out]


def near_axis_scene(seed, yaw_deg, eps):
    sc = _grid_scene(seed)
    sc.cam.csys.set_pos((eps, -eps, 0.0))
    sc.cam.csys.ryg(yaw_deg).rxg(-yaw_deg)
    return sc


def test_reference_scene_flags_no_query(oracle_lib):
    _audit(1)
    c = make_ctx(oracle_lib, build_suzanne(0), 160, 90)
    c.render(1, 3, 1, 1, 4, 0)
    n, leaf_a, leaf_b, own_a, own_b = _audit(0)
    assert n > 30000 and leaf_a == 0 and leaf_b == 0        # the reference's leaf boxes
    assert own_a == 0 and own_b == 0                        # the winner's own bounding box: holds for any structure's boxes


def test_adversarial_grid_scene_is_flagged(oracle_lib):
    _audit(1)
    _render(oracle_lib, near_axis_scene(1, 1e-6, 1e-6), 1, 129, False, 2, 2, 6)
    n, leaf_a, leaf_b, own_a, own_b = _audit(0)
    assert n > 500 and leaf_a == 0 and leaf_b >= 1          # bounce origins 9e-7 beside a face plane, winners at t = 1.1e-6
    assert own_a >= leaf_a and own_b >= leaf_b              # the tighter box is flagged whenever the leaf is
    _render(oracle_lib, near_axis_scene(1, 1e-6, 1e-6), 1, 129, False, 2, 2, 6)
    assert _audit(0) == [0, 0, 0, 0, 0]                     # switched off: counts nothing
