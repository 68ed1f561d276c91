#!/usr/bin/env python3
"""Wall time of the reference-control-flow encoder (oracle/_ref/x265_refsse or x265_refsse_2160p) on a synthetic clip, with 1
and 16 row threads. usage: time_ref_encoder.py W H FRAMES QP [SEED]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, h, n, qp = (int(a) for a in sys.argv[1:5])
seed = sys.argv[5] if len(sys.argv) > 5 else "1234"
big = h > 1088
exe = {1: os.path.join(ROOT, "oracle", "_ref", "x265_refsse_2160p" if big else "x265_refsse"),
       16: os.path.join(ROOT, "oracle", "_ref", "x265_refsse_2160p_mt" if big else "x265_refsse_mt")}
with tempfile.TemporaryDirectory() as d:
    yuv = os.path.join(d, "in.yuv")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(n), seed, yuv], check=True)
    for t in (1, 16):
        t0 = time.perf_counter()
        subprocess.run([exe[t], "-i", yuv, "-o", os.path.join(d, "o.265"), "-w", str(w), "-h", str(h), "-f", str(n), "-q", str(qp), "-ip", "100000",
                        "-sceneChange", "0", "-t", str(t)], check=True, capture_output=True)
        dt = time.perf_counter() - t0
        print(f"{w}x{h} {n} pictures QP {qp}, {t} thread(s): {dt:.2f} s, {n / dt:.2f} fps, {os.path.getsize(os.path.join(d, 'o.265'))} bytes", flush=True)
