#!/usr/bin/env python3
"""Record the intra luma mode searches of a real reference encode (TEST INFRASTRUCTURE; needs /root/reference).

`make -C oracle trace` builds oracle/_ref/x265_intra_trace (oracle/intra_trace_hook.cpp interposed on xPaddingRef,
xFilterRef, xEncIntraPred and the xSatdN table). This script encodes the 416x240 IPPP QP32 clip with it, checks the
bitstream against tests/golden/encoder_md5.json and stores a sample of the searches of every PU size in
tests/golden/intra_trace.npz: neighbouring samples before padding, availability bits, strong-smoothing flag,
source block and the SATD the encoder obtained for each of the 35 modes."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KEEP = {4: 500, 8: 400, 16: 250, 32: 90}


def main():
    exe = os.path.join(HERE, "_ref", "x265_intra_trace")
    if not os.path.exists(exe):
        sys.exit("build it first: make -C oracle trace")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))["416x240_f8_q32"]
    with tempfile.TemporaryDirectory() as d:
        yuv, out, trace = (os.path.join(d, n) for n in ("in.yuv", "out.265", "trace.bin"))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), "416", "240", "8", "1234", yuv])
        subprocess.run([exe, "-i", yuv, "-o", out, "-w", "416", "-h", "240", "-f", "8", "-q", "32", "-ip", "100",
                        "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, env=dict(os.environ, XMI_INTRA_TRACE=trace))
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == gold["md5"], "traced encode differs from the recorded bitstream"
        raw = open(trace, "rb").read()
    recs = {4: [], 8: [], 16: [], 32: []}
    pos = 0
    while pos < len(raw):
        size, valids, strong, poc = np.frombuffer(raw, np.int32, 4, pos)
        pos += 16
        ref = np.frombuffer(raw, np.uint8, 4 * size + 1, pos)
        pos += 4 * size + 1
        src = np.frombuffer(raw, np.uint8, size * size, pos).reshape(size, size)
        pos += size * size
        satd = np.frombuffer(raw, np.uint32, 35, pos)
        pos += 140
        recs[int(size)].append((int(valids), int(strong), int(poc), ref, src, satd))
    out = {}
    for n, lst in recs.items():
        step = max(1, len(lst) // KEEP[n])
        sel = lst[::step][:KEEP[n]]
        # always keep the searches without any neighbour and those of inter pictures
        sel += [r for r in lst if r[0] == 0][:8] + [r for r in lst if r[2] > 0][:40]
        t = f"n{n}_"
        out[t + "valids"] = np.array([r[0] for r in sel], np.uint8)
        out[t + "strong"] = np.array([r[1] for r in sel], np.uint8)
        out[t + "raw"] = np.stack([r[3] for r in sel])
        out[t + "src"] = np.stack([r[4] for r in sel])
        out[t + "satd"] = np.stack([r[5] for r in sel])
        print(f"{n}x{n}: {len(lst)} searches traced, {len(sel)} stored, valids seen {sorted(set(r[0] for r in lst))}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "intra_trace.npz"), **out)


if __name__ == "__main__":
    main()
