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


def run_ranks(extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--settle-ms", "0",
                          "--no-cpu-baseline", "--headline-only"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_share_this_gpu_over_gloo():
    """`bench.py --gpus 2` started by hand launches its two ranks itself; with XMI_BENCH_BACKEND=gloo both use the one GPU
    of this box: the line must carry the records of both ranks and say that the device is shared."""
    d = run_ranks({"XMI_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["gathered_records"] == 12 and len(d["config"]["rank_kernel_ms_avg"]) == 2
    assert all("shared" in x for x in d["config"]["rank_devices"])
    assert d["config"]["matching_cost_per_frame_mean"] > 0
    # every rank took its own slice of the host's cores and sized its encoders' thread groups to it
    assert d["config"]["cores_per_rank"] >= 1 and d["config"]["workers"] == max(1, d["config"]["cores_per_rank"] // 8)
    assert d["config"]["consumed_by_encoder"] is False


def test_two_ranks_over_rccl_on_two_gpus():
    """The same with the default backend (nccl = RCCL) where two devices exist: one device per rank."""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: the RCCL launch needs two")
    d = run_ranks()
    assert d["n_gpus"] == 2 and d["config"]["rank_devices"] == ["cuda:0", "cuda:1"] and d["config"]["gathered_records"] == 12
