#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output for the k_trace dispatches: tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections, json
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        kn = "k_trace" if ("k_trace" in k or "k_trace" in k) else None
        if kn: agg[kn][r["Counter_Name"]] += float(r["Counter_Value"])
for kn, c in agg.items():
    print(kn)
    for n in sorted(c): print(f"  {n:36s} {c[n]:.6g}")
    g = c.get
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print(f"  -> VALU lane utilisation  = {g('SQ_THREAD_CYCLES_VALU')/(g('SQ_ACTIVE_INST_VALU')*64)*100:.1f} %  (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64))")
    if g("SQ_WAVE_CYCLES"):
        for n in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if g(n): print(f"  -> {n}/SQ_WAVE_CYCLES = {g(n)/g('SQ_WAVE_CYCLES')*100:.1f} %")
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"): print(f"  -> VALU insts / wave = {g('SQ_INSTS_VALU')/g('SQ_WAVES'):.0f}")
print(json.dumps({k: dict(v) for k, v in agg.items()}))
