"""CPU: the reference's encoder control flow (its sources compiled where they lie into oracle/_ref by `make -C oracle
enc`) must write the recorded bitstreams (tests/golden/encoder_md5.json) with the reference's own C++/SSE4 primitives
(x265_refsse) and with every primitive taken from the oracle (x265_oracle): the oracle then stands in for the reference
inside a whole encode, including the MD5 SEI of every reconstructed picture. 416x240 at QP 22/27/32/37, a clip with a
scene cut, and 1920x1080 (the geometry the headline metric is quoted on) at QP 32 and 22."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from tests.prereq import needs_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))
SEED = {416: 1234, 1920: 7}

CASES = [("416x240_f8_q32", ("refsse", "oracle")), ("416x240_f3_q22", ("refsse", "oracle")), ("416x240_f8_q27", ("oracle",)),
         ("416x240_f8_q37", ("oracle",)), ("416x240_f8_q32_cut6_scenechange", ("refsse", "oracle")),
         ("1920x1080_f4_q32", ("refsse", "oracle")), ("1920x1080_f3_q22", ("oracle",))]


def parse(key):
    size, f, q = key.split("_")[:3]
    w, h = (int(v) for v in size.split("x"))
    cut = key.split("_cut")[1].split("_")[0] if "_cut" in key else None
    return w, h, int(f[1:]), int(q[1:]), cut


@needs_ref("x265_oracle", how="make -C oracle enc")
@pytest.mark.parametrize("key,variants", CASES, ids=[c[0] for c in CASES])
def test_recorded_bitstreams_on_cpu(tmp_path, key, variants):
    w, h, frames, qp, cut = parse(key)
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), str(SEED[w]), yuv]
                          + ([cut] if cut else []))
    want = GOLD[key]
    assert hashlib.md5(open(yuv, "rb").read()).hexdigest() == want["input_md5"]
    for v in variants:
        out = str(tmp_path / f"{v}.265")
        res = subprocess.run([os.path.join(REFDIR, f"x265_{v}"), "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(frames),
                              "-q", str(qp), "-ip", "100", "-sceneChange", "1" if cut else "0", "-t", "1"],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-1000:]
        assert (hashlib.md5(open(out, "rb").read()).hexdigest(), os.path.getsize(out)) == (want["md5"], want["bytes"]), f"x265_{v} on {key}"
