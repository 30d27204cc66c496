"""The register budget of k_trace is part of its design (80 VGPRs, 6 waves per SIMD, NO scratch - three spilled
registers cost 3.5 % in round 2, and the allocator does not keep to the budget by itself: compiling code OUT made it spill in round 3).
Cross-compiles the kernels for gfx950 (no GPU needed) and reads the assembler's resource notes of every instantiation."""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not available")
def test_no_timed_k_trace_instantiation_spills():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "isa_all.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("void k_trace<")]
    assert len(rows) == 63, len(rows)                       # 2 x 3 x 2 x 2 x 2 (+ MLDS for the render kernels, + ROCT for the three small-SAH-scene ones)
    for ln in rows:
        vg, sc, occ = (int(x) for x in re.search(r"VGPRs\s+(\d+) scratch\s+(\d+) waves/SIMD (\d+)", ln).groups())
        assert occ >= 6, ln                                  # (__launch_bounds__(256, 6))
        if ln.startswith("void k_trace<false"):
            assert sc == 0 and vg <= 80, ln                  # the timed instantiations; counting ones may spill
    head = next(ln for ln in rows if ln.startswith("void k_trace<false, 1, false, true, false, true, true>"))   # bench.py's headline
    assert "VGPRs  80 scratch    0 waves/SIMD 6" in head, head
