"""The CPU timing programs of the bench (oracle/cpu_baseline, oracle/cpu_primitives): they run, the reference's
compiled SAD table and the restatement visit the same candidates with the same checksum, and every primitive
and size of the GPU tables has a CPU row beside it."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
REF_LIB = os.path.join(ORACLE, "_ref", "libx265ref_pixel.so")


def _need(path):
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path, ROOT)} not built (python -c 'import __graft_entry__ as g; g.build()')")


def test_cpu_baseline_reference_and_port_agree(tmp_path):
    exe = os.path.join(ORACLE, "cpu_baseline")
    _need(exe)
    w, h, pad, rng_ = 128, 64, 32, 16
    rng = np.random.default_rng(5)
    cur = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref = rng.integers(0, 256, (h + 2 * pad, w + 2 * pad), dtype=np.uint8)
    path = tmp_path / "frames.bin"
    path.write_bytes(cur.tobytes() + ref.tobytes())

    def run(extra):
        out = subprocess.run([exe, str(path), str(w), str(h), str(pad), str(rng_), "2"] + extra, capture_output=True, text=True,
                             timeout=120, check=True).stdout.strip()
        return dict(x.split("=") for x in out.split())

    port = run([])
    assert port["kind"] == "port" and int(port["cands"]) == 2 * 33 * 33 * 85
    # independent numpy checksum of the same candidate list
    want = 0
    for cx in (0, 64):
        c = cur[:, cx:cx + 64].astype(np.int64)
        for dy in range(-rng_, rng_ + 1):
            for dx in range(-rng_, rng_ + 1):
                r = ref[pad + dy:pad + dy + 64, pad + cx + dx:pad + cx + dx + 64].astype(np.int64)
                want += 4 * int(np.abs(c - r).sum())      # every pixel counts once per pyramid level
    assert int(port["checksum"]) == want
    if os.path.exists(REF_LIB):
        refr = run([REF_LIB])
        assert refr["kind"] == "reference" and refr["checksum"] == port["checksum"] and refr["cands"] == port["cands"]


def test_cpu_primitives_rows():
    exe = os.path.join(ORACLE, "cpu_primitives")
    _need(exe)
    args = [exe, REF_LIB if os.path.exists(REF_LIB) else "", "0.01"]
    rows = [json.loads(x) for x in subprocess.run(args, capture_output=True, text=True, timeout=120, check=True).stdout.splitlines()]
    names = {r["cpu"] for r in rows}
    want = {f"{m} {n}x{n}" for m in ("sad", "ssd", "satd_hm") for n in (4, 8, 16, 32, 64)}
    want |= {f"satd_x264 {n}x{n}" for n in (4, 8, 16)}
    want |= {f"{m} {n}x{n}" for m in ("sub_dct", "quant", "dequant", "idct_add", "tu_chain") for n in (4, 8, 16, 32)}
    assert names == want
    assert all(r["Mblk_per_s"] > 0 for r in rows)
    if os.path.exists(REF_LIB):
        kinds = {r["cpu"]: r["kind"] for r in rows}
        assert kinds["tu_chain 32x32"] == "reference" and kinds["sad 64x64"] == "reference" and kinds["satd_hm 8x8"] == "reference"
        assert kinds["satd_x264 8x8"] == "port" and kinds["ssd 16x16"] == "port"   # NASM routines: not assembled here
