#!/usr/bin/env python3
"""Static report of the headline k_trace instantiation's code object: resources from the assembler's
kernel notes, opcode histogram, and instruction counts of the three blocks of the wave scheduler
(node step / triangle step / path block), located through the global loads that only they contain.

usage: python tools/isa_report.py [mangled-kernel-name] > profiles/r02_k_trace_isa.txt
Compiles raytracer_amd/csrc/rt_hip.hip with the product's flags to assembly (device only, no GPU needed)."""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
NAME = sys.argv[1] if len(sys.argv) > 1 else "_Z7k_traceILb0ELi1ELb0ELb1ELb0ELb1ELb1EEv9TraceArgs"   # <STATS=0, CULL=1, FIRST_HIT=0, SAH=1, DEEP=0, MLDS=1, ROCT=1>
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DRT_WAVES=6"]

with tempfile.TemporaryDirectory() as tmp:
    out = Path(tmp) / "k.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", str(ROOT / "raytracer_amd/csrc/rt_hip.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()

i = text.index("\n" + NAME + ":")
j = text.index(".end_amdhsa_kernel", i)
body = text[i:j].split("\n")
meta = text[j - 6000:j]
notes = text[text.index("\n" + NAME + ":"):]
print(f"# {NAME}")
print("# hipcc " + " ".join(FLAGS))
for key in ("NumSgprs", "NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize", "codeLenInByte"):
    m = re.search(r";\s*" + key + r":\s*(\S+)", notes)
    if m:
        print(f"{key:14s} {m.group(1)}")
for key in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr", ".amdhsa_group_segment_fixed_size", ".amdhsa_private_segment_fixed_size"):
    m = re.search(re.escape(key) + r"\s+(\S+)", text[j - 4000:j])
    if m:
        print(f"{key:40s} {m.group(1)}")


def is_inst(t):
    return bool(t) and not t.startswith((".", ";")) and not t.endswith(":")


insts = [(n, l.strip()) for n, l in enumerate(body) if is_inst(l.strip())]
ops = collections.Counter(t.split()[0] for _, t in insts)
kinds = collections.Counter("VALU" if o.startswith("v_") else "SALU" if o.startswith("s_") else "LDS" if o.startswith("ds_") else "VMEM" for _, t in insts for o in [t.split()[0]])
print(f"\n## whole kernel: {len(insts)} instructions  {dict(kinds)}")
print("flat_load/flat_store:", sum(v for k, v in ops.items() if k.startswith("flat_")), " global_load/store:", sum(v for k, v in ops.items() if k.startswith("global_")),
      " scratch_*:", sum(v for k, v in ops.items() if k.startswith("scratch_")))
pref = lambda p: sum(v for k, v in ops.items() if k.startswith(p))
print("IEEE division sequences (v_div_fixup_f32):", pref("v_div_fixup_f32"), "  v_sqrt_f32:", pref("v_sqrt_f32"), "  v_rcp_f32:", pref("v_rcp_f32"),
      "  v_mul_lo_u32 (RNG):", pref("v_mul_lo_u32"), "  s_nop:", pref("s_nop"))
print("opcode histogram (>= 8):")
for o, n in ops.most_common():
    if n >= 8:
        print(f"  {n:4d} {o}")

# ---- blocks of the scheduler: found through their loads -----------------------------------------------
labels = {l.split(":")[0]: n for n, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}


def region_around(pred, lo_pat, hi_pat):
    """instruction range of the innermost loop body that contains the first line matching pred"""
    k = next(n for n, l in enumerate(body) if pred(l))
    return k


def count(a, b):
    c = collections.Counter()
    for n in range(a, b):
        t = body[n].strip()
        if is_inst(t):
            o = t.split()[0]
            c["VALU" if o.startswith("v_") else "SALU" if o.startswith("s_") else "LDS" if o.startswith("ds_") else "VMEM"] += 1
    return dict(c)


# four consecutive global_load_dwordx4 from one base register = a 64-byte record fetch: the first such group inside the
# traversal loop is the triangle record (tri_geo), the second the node record
groups = []
n = 0
while n < len(body):
    if "global_load_dwordx4" in body[n]:
        g = [n]
        m = n + 1
        while m < len(body) and len(g) < 4 and (m - g[-1]) <= 3:
            if "global_load_dwordx4" in body[m]:
                g.append(m)
            m += 1
        if len(g) == 4:
            groups.append(g[0])
            n = g[-1]
    n += 1
lab_sorted = sorted(labels.items(), key=lambda kv: kv[1])


def enclosing(k, back_label_count):
    prev = [p for _, p in lab_sorted if p <= k]
    return prev[-1] if prev else 0


print("\n## scheduler blocks (static counts of the straight-line path through each block, every divergent region included once)")
if len(groups) >= 2:
    tri_k, node_k = groups[0], groups[1]
    # block starts: the loop-header label before the fetch that is preceded by a vote (s_bcnt1) ...; simplest robust
    # delimitation: from the previous label carrying "Loop Header" / "in Loop" that follows an s_cbranch_vcc* to the
    # next s_branch back to the vote
    def block(k):
        a = k
        while a > 0 and not ("s_bcnt1_i32_b64" in body[a] or "Loop Header" in body[a]):
            a -= 1
        b = k
        while b < len(body) and not body[b].strip().startswith("s_branch"):
            b += 1
        return a, b
    for name, k in (("triangle step (one triangle of a leaf per lane)", tri_k), ("node step (one interior record = two child boxes per lane)", node_k)):
        a, b = block(k)
        print(f"{name}: asm lines {a}..{b}: {count(a, b)}")
    print("  (the node step's range holds BOTH forms of the slab test: the min/max form every wave with finite 1/d takes")
    print("   and the reference's compare-and-swap ladder for a wave with an exactly zero direction component)")
    a, b = block(node_k)
    print(f"path block + sample hand-out + query start (everything after the traversal loop): {count(b, len(body))}")
print("\nDynamic counts per executed block, fitted from SQ_INSTS_VALU / SQ_INSTS_SALU against the wave-level step counters (tools/fit_steps.sh, DESIGN.md 8),")
print("BEFORE the lane-mask bookkeeping of round 2 (after it the three-threshold fit is ill-conditioned; per launch: see the counters below):")
print("  node step 131 VALU + 71 SALU, triangle step 89 + 60, path block 1097 + 368 (SAH structure, culled)")
