"""bench.py prints ONE JSON line with the keys the driver's contract names (metric, value, unit, n_gpus,
steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus the
`roofline` and `cpu_baseline` objects; run here in a child process on one GPU with the smallest step count."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_line_contract():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-passes", "1"],
                       capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1000 and 1.0 < d["ms_per_step"] < 1000.0
    # value = rays of the timed steps / their wall time
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) / d["value"] < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    h = r["hbm"]
    # algorithmic = 24 B per pixel per launch (accumulator); the staging stream is design overhead, named as such
    assert h["algorithmic_bytes_per_launch"] == 24 * 1920 * 1080 and h["design_overhead_bytes_per_launch"] == 24 * 1920 * 1080 * 64
    assert r["traffic"] is None or r["traffic"] >= h["algorithmic_bytes_per_launch"]
    assert r["traffic"] is None or abs(h["traffic_over_algorithmic"] - r["traffic"] / h["algorithmic_bytes_per_launch"]) < 0.01
    # the headline's evidence for THIS frame, and the fast mode with the weaker assumption, ride on the same line
    assert d["cull_audit_flagged"] == 0 and d["cull_audit"]["queries_won_by_a_triangle"] > 5e8
    assert d["sah_rule_a_only"]["ms_per_step"] > 0.9 * d["ms_per_step"]
    assert d["reference_structure"]["culled"]["ms_per_step"] > d["ms_per_step"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
