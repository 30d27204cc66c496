#!/usr/bin/env python3
"""Copy the output of tools/prof_bench.sh (gpurun_out/<dir>) into profiles/ for one workload and refresh its entry in
profiles/pmc_latest.json (what bench.py reports as roofline.traffic):
    python tools/update_profiles.py <round tag, e.g. r02> <workload> <gpurun_out dir> "<commit note>" """
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag, wl, src, note = sys.argv[1], sys.argv[2], Path(sys.argv[3]), sys.argv[4]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

W, H, P = (bench.WORKLOADS[wl][k] for k in ("width", "height", "passes"))
prof = ROOT / "profiles"
shutil.copy(src / "trace" / "trace_kernel_stats.csv", prof / f"{tag}_{wl}_kernel_stats.csv")
shutil.copy(src / "summary.txt", prof / f"{tag}_{wl}_rocprof_summary.txt")
pmc = json.loads((src / "pmc.json").read_text())
name, k = next((n, v) for n, v in pmc["kernels"].items() if n.startswith("void k_trace<false"))
avg = calls = None
for row in csv.DictReader(open(src / "trace" / "trace_kernel_stats.csv")):
    if row["Name"].startswith(name):
        avg, calls = float(row["AverageNs"]) / 1e6, int(row["Calls"])
latest_p = prof / "pmc_latest.json"
latest = json.loads(latest_p.read_text()) if latest_p.exists() else {}
old = latest.get(wl, {})
latest[wl] = {
    "hbm_bytes_per_launch": int(k["hbm_bytes_per_launch"]),
    "kernel": name.replace("void ", ""),
    "FETCH_SIZE_KB": k["FETCH_SIZE"],
    "WRITE_SIZE_KB": k["WRITE_SIZE"],
    "l2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]),
    "correction": old.get("correction", "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM); WRITE_SIZE as read; both x1024. "
                                        "FETCH_SIZE counts L2->fabric reads: what the Infinity Cache serves is included"),
    "file": f"profiles/{tag}_{wl}_rocprof_summary.txt",
    "commit": note,
    "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) -- " + pmc["command"],
    # the accumulator, 12 B read + 12 B written per pixel per launch; the staging stream is design overhead
    "algorithmic_hbm_bytes_per_launch": 24 * W * H,
    "design_overhead_bytes_per_launch": 24 * W * H * P,
    "avg_launch_ms_rocprof": avg,
    "launches_traced": calls,
}
latest_p.write_text(json.dumps(latest, indent=1) + "\n")
print(wl, latest[wl]["hbm_bytes_per_launch"], avg, calls)
