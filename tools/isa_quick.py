#!/usr/bin/env python3
"""Registers, scratch and opcode-class counts of one k_trace instantiation for a set of hipcc flags (no GPU needed).
usage: python tools/isa_quick.py [-DFLAG ...]  (RT_WAVES defaults to 6; RT_KERNEL=<mangled name> picks another instantiation)"""
import collections
import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
NAME = os.environ.get("RT_KERNEL", "_Z7k_traceILb0ELi1ELb0ELb1ELb0ELb1ELb1EEv9TraceArgs")   # <STATS=0, CULL=1, FIRST_HIT=0, SAH=1, DEEP=0, MLDS=1, ROCT=1>
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"] + sys.argv[1:]
if not any(f.startswith("-DRT_WAVES") for f in flags):
    flags.append("-DRT_WAVES=6")
with tempfile.TemporaryDirectory() as tmp:
    out = Path(tmp) / "k.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", str(ROOT / "raytracer_amd/csrc/rt_hip.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    t = out.read_text()
i = t.index("\n" + NAME + ":")
j = t.index(".end_amdhsa_kernel", i)
notes = t[i:]
r = {k: re.search(r";\s*" + k + r":\s*(\S+)", notes).group(1) for k in ("NumVgprs", "ScratchSize", "Occupancy")}
c = collections.Counter()
for l in t[i:j].split("\n"):
    s = l.strip()
    if s and not s.startswith((".", ";")) and not s.endswith(":"):
        c[s.split()[0]] += 1
cls = lambda p: sum(v for k, v in c.items() if k.startswith(p))
print(" ".join(sys.argv[1:]) or "(product flags)", r, "VALU", cls("v_"), "SALU", cls("s_"), "flat", cls("flat_"), "global", cls("global_"), "ds", cls("ds_"),
      "scratch", cls("scratch_"), "v_mov", cls("v_mov"))
