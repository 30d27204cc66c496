"""Effect of the one arithmetic choice that departs from "true division" (DESIGN.md 2): the
slab test uses (b-o)*(1/d).  The oracle can run both; the images must agree."""
import os

import numpy as np

from conftest import build_suzanne, make_ctx


def test_reciprocal_vs_true_division(oracle_lib):
    res = {}
    for mode in ("0", "1"):
        os.environ["RT_ORACLE_TRUE_DIV"] = mode          # read by rt_create
        try:
            c = make_ctx(oracle_lib, build_suzanne(0), 320, 180)
        finally:
            os.environ.pop("RT_ORACLE_TRUE_DIV", None)
        ids, ts = c.first_hit(1)
        c.render(1, 1, 1, 1, 4, 0)
        res[mode] = (ids, ts, c.read_mean())
    n_ids = int((res["0"][0] != res["1"][0]).sum())
    n_rad = int((res["0"][2] != res["1"][2]).any(axis=2).sum())
    print(f"first-hit ids differing: {n_ids} / {320*180}; radiance pixels differing: {n_rad}")
    assert n_ids == 0
    assert n_rad <= 5          # 5 bounces per path: a box-edge graze can flip one path
