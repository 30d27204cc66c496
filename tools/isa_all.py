#!/usr/bin/env python3
"""VGPRs, scratch and occupancy of EVERY k_trace instantiation (and the prep kernels) with the product's flags: the
kernel's budget is 80 VGPRs / no scratch at 6 waves per SIMD, and the register allocator does not keep to
it by itself.  usage: python tools/isa_all.py [extra hipcc flags]   (exit code 1 if a non-counting instantiation spills)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DRT_WAVES=6"] + sys.argv[1:]
with tempfile.TemporaryDirectory() as tmp:
    out = Path(tmp) / "k.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", str(ROOT / "raytracer_amd/csrc/rt_hip.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    t = out.read_text()
bad = 0
for m in re.finditer(r"^(_Z\w+):\s*; @", t, re.M):
    name = m.group(1)
    notes = t[m.start():]
    r = {k: re.search(r";\s*" + k + r":\s*(\S+)", notes).group(1) for k in ("NumVgprs", "ScratchSize", "Occupancy")}
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(TraceArgs)", "")
    spill = int(r["ScratchSize"]) > 0
    counting = dem.startswith("void k_trace<true")
    if spill and not counting:
        bad += 1
    print(f"{dem:60s} VGPRs {r['NumVgprs']:>3s} scratch {r['ScratchSize']:>4s} waves/SIMD {r['Occupancy']}" + ("   <-- spills" if spill else ""))
print(f"{bad} timed instantiation(s) with scratch")
sys.exit(1 if bad else 0)
