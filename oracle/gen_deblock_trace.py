#!/usr/bin/env python3
"""Record the in-loop deblocking of a real reference encode (TEST INFRASTRUCTURE; needs /root/reference).

`make -C oracle trace` builds oracle/_ref/x265_deblock_trace (oracle/deblock_trace_hook.cpp interposed on xDeblockFrame).
This script encodes the 416x240 IPPP clip at QP 37 (strong filtering) with it, checks the bitstream against
tests/golden/encoder_md5.json and stores the I picture and two P pictures in tests/golden/deblock_trace.npz: planes
before and after, boundary strengths, QP table, offsets."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KEEP = (0, 2, 6)


def main():
    exe = os.path.join(HERE, "_ref", "x265_deblock_trace")
    if not os.path.exists(exe):
        sys.exit("build it first: make -C oracle trace")
    qp = 37
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f8_q{qp}"]
    with tempfile.TemporaryDirectory() as d:
        yuv, out, tr = (os.path.join(d, n) for n in ("in.yuv", "out.265", "trace.bin"))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), "416", "240", "8", "1234", yuv])
        subprocess.run([exe, "-i", yuv, "-o", out, "-w", "416", "-h", "240", "-f", "8", "-q", str(qp), "-ip", "100",
                        "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, env=dict(os.environ, XMI_DEBLOCK_TRACE=tr))
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == gold["md5"], "traced encode differs from the recorded bitstream"
        raw = open(tr, "rb").read()
    pos, out = 0, {}
    pocs = []
    while pos < len(raw):
        w, h, poc, slice_, tco, bo, cb, cr = (int(v) for v in np.frombuffer(raw, np.int32, 8, pos))
        pos += 32

        def take(n, shape):
            nonlocal pos
            a = np.frombuffer(raw, np.uint8, n, pos).reshape(shape)
            pos += n
            return a
        rec = {}
        for tag in ("y0", "u0", "v0"):
            rec[tag] = take(w * h if tag == "y0" else w * h // 4, (h, w) if tag == "y0" else (h // 2, w // 2))
        rec["vbs"], rec["hbs"] = take(w * h // 16, (-1,)), take(w * h // 16, (-1,))
        rec["qp"] = take(w * h // 16, (h // 4, w // 4))
        for tag in ("y1", "u1", "v1"):
            rec[tag] = take(w * h if tag == "y1" else w * h // 4, (h, w) if tag == "y1" else (h // 2, w // 2))
        changed = int((rec["y0"] != rec["y1"]).sum())
        print(f"poc {poc}: slice {slice_}, {w}x{h}, bs=1: {int((rec['vbs'] == 1).sum() + (rec['hbs'] == 1).sum())}, "
              f"bs=2: {int((rec['vbs'] == 2).sum() + (rec['hbs'] == 2).sum())}, luma samples changed {changed}, "
              f"chroma {int((rec['u0'] != rec['u1']).sum() + (rec['v0'] != rec['v1']).sum())}")
        if poc in KEEP:
            pocs.append(poc)
            out[f"p{poc}_meta"] = np.array([w, h, slice_, tco, bo, cb, cr], np.int32)
            for k, a in rec.items():
                out[f"p{poc}_{k}"] = a
    out["pocs"] = np.array(pocs, np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "deblock_trace.npz"), **out)


if __name__ == "__main__":
    main()
