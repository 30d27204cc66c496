"""Audit of the distance-culling rules on the bench workloads (CPU, oracle/rt_oracle.c: rto_cull_audit): for every
closest-hit query whose winner is a triangle, could rule (a) (entry beyond the best hit) or rule (b) (exit before
5e-7) skip the winner's leaf box in the reference's tree, or the winner's own bounding box (which lies inside every
box that any structure over the same triangles puts it in: covers RT_ACCEL_SAH)?  Independent of traversal order; zero flagged queries = a culling traversal cannot
lose the reference's winner on these rays.   python tools/cull_audit.py [workload ...] > profiles/rNN_cull_audit.txt"""
import ctypes
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import WORKLOADS, build_scene  # noqa: E402
from raytracer_amd.ffi import RtLib  # noqa: E402

SO = ROOT / "oracle" / "librt_oracle.so"
# (width, height, passes) audited per workload: bounded CPU time, same scene / camera / seeds as the bench
SAMPLE = {"suzanne1080": (1920, 1080, 64), "dragon": (1920, 1080, 16), "subdiv4k": (1920, 1080, 4)}


def main():
    names = sys.argv[1:] or list(SAMPLE)
    audit = ctypes.CDLL(str(SO)).rto_cull_audit
    out = (ctypes.c_ulonglong * 5)()
    lib = RtLib(SO)
    print("# workload | image, passes, bounces | queries with a triangle winner | leaf box: rule (a) | rule (b) | own box: rule (a) | rule (b) | wall s")
    for name in names:
        wl = WORKLOADS[name]
        w, h, passes = SAMPLE[name]
        sc = build_scene(wl["scene"], lib)
        c = lib.create()
        c.load_scene(sc)
        c.load_camera(sc.cam)
        c.resize(w, h)
        audit(ctypes.c_int(1), None)
        t0 = time.time()
        c.render(1, passes, 1, wl["rpp"], wl["bounces"], 0)
        audit(ctypes.c_int(0), out)
        print(f"{name} | {w}x{h}, passes 1..{passes}, {wl['bounces']} bounces | {int(out[0])} | {int(out[1])} | {int(out[2])} | {int(out[3])} | {int(out[4])} | {time.time() - t0:.0f}", flush=True)


if __name__ == "__main__":
    main()
