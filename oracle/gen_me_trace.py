#!/usr/bin/env python3
"""Record the motion searches of a real reference encode (TEST INFRASTRUCTURE; needs /root/reference).

`make -C oracle trace` builds oracle/_ref/x265_trace: the reference's CLI and encoder sources with
oracle/me_trace_hook.cpp interposed on xMotionSearch. This script encodes the 416x240 IPPP QP32 clip of
tests/golden/encoder_md5.json with it, checks that the bitstream is the recorded one, and stores the source luma,
the padded reference luma and every xMotionSearch call (inputs and outputs) of a few P pictures in
tests/golden/me_trace.npz."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KEEP_POCS = (1, 4, 7)


def main():
    exe = os.path.join(HERE, "_ref", "x265_trace")
    if not os.path.exists(exe):
        sys.exit("build it first: make -C oracle trace")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))["416x240_f8_q32"]
    with tempfile.TemporaryDirectory() as d:
        yuv, out, trace = (os.path.join(d, n) for n in ("in.yuv", "out.265", "trace.bin"))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), "416", "240", "8", "1234", yuv])
        subprocess.run([exe, "-i", yuv, "-o", out, "-w", "416", "-h", "240", "-f", "8", "-q", "32", "-ip", "100",
                        "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, env=dict(os.environ, XMI_ME_TRACE=trace))
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == gold["md5"], "traced encode differs from the recorded bitstream"
        raw = open(trace, "rb").read()
    pos, frames, calls = 0, {}, {}
    while pos < len(raw):
        head = np.frombuffer(raw, np.int32, 1, pos)[0]
        if head == -1:
            _, poc, w, h, pad = np.frombuffer(raw, np.int32, 5, pos)
            pos += 20
            cur = np.frombuffer(raw, np.uint8, w * h, pos).reshape(h, w)
            pos += w * h
            ref = np.frombuffer(raw, np.uint8, (w + 2 * pad) * (h + 2 * pad), pos).reshape(h + 2 * pad, w + 2 * pad)
            pos += ref.size
            frames[int(poc)] = (cur, ref, int(pad))
            calls[int(poc)] = []
        else:
            rec = np.frombuffer(raw, np.int32, 16, pos)
            lam = np.frombuffer(raw, np.float64, 1, pos + 64)[0]
            cost, sad = np.frombuffer(raw, np.uint32, 2, pos + 72)
            pos += 80
            calls[int(rec[0])].append((rec.copy(), lam, cost, sad))
    out = {"pocs": np.array(KEEP_POCS, np.int32)}
    total = 0
    for poc in KEEP_POCS:
        cur, ref, pad = frames[poc]
        c = calls[poc]
        out[f"p{poc}_cur"], out[f"p{poc}_ref"] = cur, ref
        out[f"p{poc}_req"] = np.stack([r[0][1:13] for r in c]).astype(np.int32)     # x y size mvp0 mvp1 zero min max
        out[f"p{poc}_lambda"] = np.array([r[1] for r in c], np.float64)
        out[f"p{poc}_res"] = np.array([[r[0][13], r[0][14], r[2], r[3], r[0][15]] for r in c], np.int64)  # mvx mvy cost sad mvp_idx
        out["pad"] = np.array([pad], np.int32)
        total += len(c)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "me_trace.npz"), **out)
    print(f"{sum(len(v) for v in calls.values())} searches in {len(calls)} pictures traced; {total} of pictures {KEEP_POCS} stored")


if __name__ == "__main__":
    main()
