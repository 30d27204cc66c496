#!/usr/bin/env python3
"""Condense tools/prof_bench.sh output: per-kernel stats table, every PMC counter per k_trace
instantiation (mean per launch), and the HBM bytes per launch of the timed instantiation.

FETCH_SIZE / WRITE_SIZE are in kilobytes (x1024); the gfx950 correction of MI355X_MICROARCH.md
"HBM" applies to FETCH_SIZE: x2 (128-byte requests are tallied at 64 bytes).
Writes <out>/summary.txt and <out>/pmc.json."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
lines = []
cmd = ""
try:
    cmd = open(f"{out}/command.txt").read().strip()
    lines.append(f"# command: {cmd}")
except OSError:
    pass
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    lines.append(f"# {f.split(out)[-1]}")
    lines += [l.rstrip() for l in open(f)]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trace" not in k:
            continue
        per[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
js = {"command": cmd, "kernels": {}}
for k, c in sorted(per.items()):
    lines.append(f"# PMC {k}")
    for n, v in sorted(c.items()):
        lines.append(f"  {n:34s} launches={len(v)} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        lines.append(f"  -> HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {m['hbm_bytes_per_launch']:.6g}")
    if m.get("SQ_ACTIVE_INST_VALU") and m.get("SQ_THREAD_CYCLES_VALU"):
        lines.append(f"  -> VALU lane utilisation = {m['SQ_THREAD_CYCLES_VALU'] / (m['SQ_ACTIVE_INST_VALU'] * 64) * 100:.1f} %")
    if m.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if m.get(n):
                lines.append(f"  -> {n} / SQ_WAVE_CYCLES = {m[n] / m['SQ_WAVE_CYCLES'] * 100:.1f} %")
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) > 0:
        lines.append(f"  -> L2 hit rate = {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']) * 100:.2f} %")
    js["kernels"][k] = m
open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
open(f"{out}/pmc.json", "w").write(json.dumps(js, indent=1))
print("\n".join(lines))
