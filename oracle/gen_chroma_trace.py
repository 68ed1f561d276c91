#!/usr/bin/env python3
"""Record the intra chroma mode searches of real reference encodes (TEST INFRASTRUCTURE; needs /root/reference).

`make -C oracle trace` builds oracle/_ref/x265_chroma_trace (oracle/chroma_trace_hook.cpp interposed on xEncodeChromaCU,
xEncIntraPred and the xSatdN table). This script encodes the 416x240 clip at QP 22 and QP 32 with it, checks the
bitstreams against tests/golden/encoder_md5.json and stores the searches of the I pictures in
tests/golden/chroma_trace.npz: per search the size, the five candidate modes, the padded neighbouring samples and source
blocks of U and V, the ten SATDs and the candidate index the encoder kept."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RUNS = ((22, 3, 400), (32, 8, 160))      # qp, frames, searches kept at most


def main():
    exe = os.path.join(HERE, "_ref", "x265_chroma_trace")
    if not os.path.exists(exe):
        sys.exit("build it first: make -C oracle trace")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))
    recs = {4: [], 8: [], 16: [], 32: []}
    for qp, frames, keep in RUNS:
        with tempfile.TemporaryDirectory() as d:
            yuv, bits, tr = (os.path.join(d, n) for n in ("in.yuv", "out.265", "trace.bin"))
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), "416", "240", str(frames), "1234", yuv])
            subprocess.run([exe, "-i", yuv, "-o", bits, "-w", "416", "-h", "240", "-f", str(frames), "-q", str(qp), "-ip", "100",
                            "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, env=dict(os.environ, XMI_CHROMA_TRACE=tr))
            assert hashlib.md5(open(bits, "rb").read()).hexdigest() == gold[f"416x240_f{frames}_q{qp}"]["md5"], "traced encode differs"
            raw = open(tr, "rb").read()
        pos = n = 0
        while pos < len(raw):
            hdr = np.frombuffer(raw, np.int32, 8, pos)
            pos += 32
            s = int(hdr[0])
            parts = []
            for count, dt in ((4 * s + 1, np.uint8), (4 * s + 1, np.uint8), (s * s, np.uint8), (s * s, np.uint8), (5, np.uint32), (5, np.uint32)):
                parts.append(np.frombuffer(raw, dt, count, pos).copy())
                pos += count * np.dtype(dt).itemsize
            if n < keep:
                recs[s].append((hdr[2:8].copy(), *parts))
            n += 1
        print(f"qp {qp}: {n} searches")
    out = {}
    for s, rs in recs.items():
        if not rs:
            continue
        out[f"n{s}_best_modes"] = np.stack([r[0] for r in rs]).astype(np.int32)      # best index, then the five modes
        out[f"n{s}_ref_u"], out[f"n{s}_ref_v"] = np.stack([r[1] for r in rs]), np.stack([r[2] for r in rs])
        out[f"n{s}_src_u"] = np.stack([r[3] for r in rs]).reshape(-1, s, s)
        out[f"n{s}_src_v"] = np.stack([r[4] for r in rs]).reshape(-1, s, s)
        out[f"n{s}_satd_u"], out[f"n{s}_satd_v"] = np.stack([r[5] for r in rs]), np.stack([r[6] for r in rs])
        print(f"size {s}: {len(rs)} searches, kept candidate histogram {np.bincount(out[f'n{s}_best_modes'][:, 0], minlength=5)}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chroma_trace.npz"), **out)


if __name__ == "__main__":
    main()
