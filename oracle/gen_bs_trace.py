#!/usr/bin/env python3
"""Record how real reference encodes derive their deblocking boundary strengths (TEST INFRASTRUCTURE; needs /root/reference).

`make -C oracle trace` builds oracle/_ref/x265_bs_trace (oracle/bs_trace_hook.cpp interposed on
ff_hevc_deblocking_boundary_strengths and xDeblockFrame). This script encodes the 416x240 IPPP clip at QP 32 and QP 22
with it, checks the bitstreams against tests/golden/encoder_md5.json and stores, for an I picture and P pictures of
both, the picture-level maps (unit size and luma cbf per 4x4, motion vector per 8x8) and the vertical_bs / horizontal_bs
arrays the encoder had when the picture was complete: tests/golden/bs_trace.npz."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RUNS = ((32, 8, (0, 1, 5)), (22, 3, (0, 1, 2)))      # qp, frames, pictures kept


def main():
    exe = os.path.join(HERE, "_ref", "x265_bs_trace")
    if not os.path.exists(exe):
        sys.exit("build it first: make -C oracle trace")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))
    out, tags = {}, []
    for qp, frames, keep in RUNS:
        with tempfile.TemporaryDirectory() as d:
            yuv, bits, tr = (os.path.join(d, n) for n in ("in.yuv", "out.265", "trace.bin"))
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), "416", "240", str(frames), "1234", yuv])
            subprocess.run([exe, "-i", yuv, "-o", bits, "-w", "416", "-h", "240", "-f", str(frames), "-q", str(qp), "-ip", "100",
                            "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, env=dict(os.environ, XMI_BS_TRACE=tr))
            assert hashlib.md5(open(bits, "rb").read()).hexdigest() == gold[f"416x240_f{frames}_q{qp}"]["md5"], "traced encode differs"
            raw = open(tr, "rb").read()
        pos = 0
        while pos < len(raw):
            w, h, poc, slice_ = (int(v) for v in np.frombuffer(raw, np.int32, 4, pos))
            pos += 16
            n4, n8 = (w // 4) * (h // 4), (w // 8) * (h // 8)
            lg = np.frombuffer(raw, np.uint8, n4, pos).reshape(h // 4, w // 4)
            cbf = np.frombuffer(raw, np.uint8, n4, pos + n4).reshape(h // 4, w // 4)
            mv = np.frombuffer(raw, np.int16, 2 * n8, pos + 2 * n4).reshape(h // 8, w // 8, 2)
            pos += 2 * n4 + 4 * n8
            vbs, hbs = np.frombuffer(raw, np.uint8, n4, pos), np.frombuffer(raw, np.uint8, n4, pos + n4)
            pos += 2 * n4
            used = (w // 8) * (h // 4)
            assert not vbs[used:].any() and not hbs[used:].any()
            print(f"qp {qp} poc {poc}: slice {slice_}, {w}x{h}, unit sizes {sorted(set(1 << int(v) for v in np.unique(lg)))}, "
                  f"cbf set on {int(cbf.sum())} 4x4, vertical bs {np.bincount(vbs[:used], minlength=3)}, horizontal {np.bincount(hbs[:used], minlength=3)}")
            if poc in keep:
                t = f"q{qp}_p{poc}"
                tags.append(t)
                out[f"{t}_meta"] = np.array([w, h, slice_], np.int32)
                out[f"{t}_unit_log2"], out[f"{t}_cbf"], out[f"{t}_mv"] = lg, cbf, mv
                out[f"{t}_vbs"], out[f"{t}_hbs"] = vbs[:used].copy(), hbs[:used].copy()
    out["tags"] = np.array(tags)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bs_trace.npz"), **out)


if __name__ == "__main__":
    main()
