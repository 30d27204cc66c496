#!/bin/bash
# VALU instructions per node / triangle / path step: three refill thresholds give three equations
# (SQ_INSTS_VALU of one 8-pass launch, and the wave-level step counts of the same work).
set -u
OUT=${1:-gpurun_out/fit}; mkdir -p "$OUT"; export TMPDIR=/tmp
for r in 8 16 40; do
  RT_REFILL_BELOW=$r RT_CULL=1 RT_STATS=1 timeout -k 10 60 python3 tools/prof_run.py 8 > "$OUT/stats_$r.txt" 2>&1 || exit 1
  RT_REFILL_BELOW=$r RT_CULL=1 timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_$r" -o p -- python3 tools/prof_run.py 8 > "$OUT/pmc_$r.log" 2>&1 || exit 1
done
python3 - "$OUT" <<'PY'
import sys, re, csv, glob
import numpy as np
out = sys.argv[1]
A, b, bs = [], [], []
for r in (8, 16, 40):
    t = open(f"{out}/stats_{r}.txt").read()
    rays = int(re.search(r"rays=(\d+)", t).group(1))
    m = re.search(r"node steps/ray=([\d.]+) avg lanes=([\d.]+); tri steps/ray=([\d.]+) avg lanes=([\d.]+); path steps/ray=([\d.]+) avg lanes=([\d.]+)", t)
    ns, nl, ts, tl, ps, pl = (float(x) for x in m.groups())
    row = [ns / nl * rays, ts / tl * rays, ps / pl * rays]
    valu = salu = 0.0
    for f in glob.glob(f"{out}/pmc_{r}/**/*counter_collection.csv", recursive=True):
        for q in csv.DictReader(open(f)):
            if "k_trace" in q["Kernel_Name"]:
                if q["Counter_Name"] == "SQ_INSTS_VALU": valu += float(q["Counter_Value"])
                if q["Counter_Name"] == "SQ_INSTS_SALU": salu += float(q["Counter_Value"])
    print(f"refill {r}: wave steps node {row[0]:.3g} tri {row[1]:.3g} path {row[2]:.3g}; VALU {valu:.4g} SALU {salu:.4g}")
    A.append(row); b.append(valu); bs.append(salu)
x = np.linalg.solve(np.array(A), np.array(b)); y = np.linalg.solve(np.array(A), np.array(bs))
print(f"VALU per step: node {x[0]:.0f}, triangle {x[1]:.0f}, path {x[2]:.0f};  SALU per step: node {y[0]:.0f}, triangle {y[1]:.0f}, path {y[2]:.0f}")
PY
