"""GPU: bench.py prints one JSON line with the agreed keys, and its figures hang together."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_line():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "satd"):
        assert key in d, key
    assert d["unit"] == "Gcand/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "1920x1080" in d["config"]["workload"]
    # value = candidates of all steps / wall time
    cands = d["config"]["cands_per_step"]
    assert cands == 510 * 129 * 129 * 85
    assert abs(d["value"] - cands / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["algorithmic_bytes_per_launch"] == 1920 * 1088 + 2080 * 1248 + 510 * 129 * 129 * 180
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.2 < r["frac"] < 1.0
    assert r["kernel_ms_avg"] <= d["ms_per_step"] * 1.05          # the kernel is what a step consists of
    assert r["traffic"] is None or r["traffic"] >= 0.95 * r["algorithmic_bytes_per_launch"]
    for name, row in d["satd"].items():
        assert 0.2 < row["frac"] < 1.0, (name, row)
