#!/usr/bin/env python3
"""Record the transform-unit chains of a real reference encode (TEST INFRASTRUCTURE; needs /root/reference).

`make -C oracle trace` builds oracle/_ref/x265_tu_trace (oracle/tu_trace_hook.cpp interposed on xSubDct, xQuant,
xDeQuant and xIDctAdd). This script encodes the 416x240 IPPP clip with it at QP 32 and QP 22, checks each bitstream
against tests/golden/encoder_md5.json and stores a sample of the chains of every TU size in tests/golden/tu_trace.npz:
source and prediction block, DST flag, QP, slice type, the levels and their sum as the encoder got them, and the
reconstructed block (the prediction when every level is zero)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KEEP = {4: 700, 8: 500, 16: 260, 32: 90}


def trace(qp, frames):
    exe = os.path.join(HERE, "_ref", "x265_tu_trace")
    if not os.path.exists(exe):
        sys.exit("build it first: make -C oracle trace")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    with tempfile.TemporaryDirectory() as d:
        yuv, out, tr = (os.path.join(d, n) for n in ("in.yuv", "out.265", "trace.bin"))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), "416", "240", str(frames), "1234", yuv])
        subprocess.run([exe, "-i", yuv, "-o", out, "-w", "416", "-h", "240", "-f", str(frames), "-q", str(qp), "-ip", "100",
                        "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, env=dict(os.environ, XMI_TU_TRACE=tr))
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == gold["md5"], "traced encode differs from the recorded bitstream"
        raw = open(tr, "rb").read()
    recs = []
    pos = 0
    while pos < len(raw):
        size, dst4, q, slice_, poc, stride, total, has_rec = (int(v) for v in np.frombuffer(raw, np.int32, 8, pos))
        pos += 32
        n2 = size * size
        pix = np.frombuffer(raw, np.uint8, n2, pos).reshape(size, size)
        pred = np.frombuffer(raw, np.uint8, n2, pos + n2).reshape(size, size)
        level = np.frombuffer(raw, np.int16, n2, pos + 2 * n2).reshape(size, size)
        rec = np.frombuffer(raw, np.uint8, n2, pos + 4 * n2).reshape(size, size)
        pos += 5 * n2
        recs.append((size, dst4, q, slice_, poc, stride, total, has_rec, pix, pred, level, rec))
    return recs


def main():
    recs = trace(32, 8) + trace(22, 3)
    out = {}
    for n in (4, 8, 16, 32):
        lst = [r for r in recs if r[0] == n]
        coded = [r for r in lst if r[6]]
        zero = [r for r in lst if not r[6]]
        step = max(1, len(coded) // KEEP[n])
        sel = coded[::step][:KEEP[n]] + zero[::max(1, len(zero) // (KEEP[n] // 4))][:KEEP[n] // 4]
        t = f"n{n}_"
        out[t + "meta"] = np.array([r[1:8] for r in sel], np.int32)     # dst4, qp, slice, poc, stride, sum, has_rec
        out[t + "pix"] = np.stack([r[8] for r in sel])
        out[t + "pred"] = np.stack([r[9] for r in sel])
        out[t + "level"] = np.stack([r[10] for r in sel])
        out[t + "rec"] = np.stack([r[11] for r in sel])
        m = out[t + "meta"]
        print(f"{n}x{n}: {len(lst)} chains traced ({len(coded)} with levels), {len(sel)} stored; qps {sorted(set(m[:, 1]))}, "
              f"slices {sorted(set(m[:, 2]))}, strides {sorted(set(m[:, 4]))}, dst {int(m[:, 0].sum())}, without reconstruction {int((m[:, 6] == 0).sum())}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tu_trace.npz"), **out)


if __name__ == "__main__":
    main()
