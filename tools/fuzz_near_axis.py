"""Exploration (GPU): rays whose direction has a tiny but non-zero component, travelling a rounding error beside
box faces of grid-aligned scenes.  The slab values on that axis are huge (|1/d| up to 1e7) and the culling rules
see them.  Every mode of the HIP library against the oracle; prints the cases that differ.
  python tools/fuzz_near_axis.py [n_seeds]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np

from raytracer_amd.ffi import RtLib
from test_gpu_fuzz import _grid_scene, _render

o = RtLib(str(ROOT / "oracle" / "librt_oracle.so"))
g = RtLib()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = total = 0
for seed in range(n_seeds):
    for yaw in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):          # degrees
        for eps in (0.0, 1e-7, -3e-7, 1e-6):
            for (w, h) in ((1, 129), (129, 1)):
                def scene():
                    sc = _grid_scene(seed)
                    sc.cam.csys.set_pos((eps, -eps, 0.0))
                    sc.cam.csys.ryg(yaw).rxg(-yaw)
                    return sc
                io, to, a, so = _render(o, scene(), w, h, False, 2, 2, 6)
                for sah in (False, True):
                    for cull in (False, True):
                        ig, tg, b, sg = _render(g, scene(), w, h, cull, 2, 2, 6, sah=sah)
                        total += 1
                        nid = int((io != ig).sum())
                        npx = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
                        if nid or npx:
                            bad += 1
                            print(f"seed {seed} yaw {yaw} eps {eps} {w}x{h} sah={sah} cull={cull}: {nid} ids, {npx} pixels differ", flush=True)
    print(f"# seed {seed} done, {bad} of {total} renders differ so far", flush=True)
print(f"# {bad} of {total} renders differ")
