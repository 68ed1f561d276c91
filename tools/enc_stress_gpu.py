#!/usr/bin/env python3
"""GPU box: streams of the library's encoder against the reference-control-flow build (oracle/_ref/x265_refsse*) on clips the
test suite does not hold - other sizes (also sizes that are no multiple of 64), QPs, seeds, lengths, a scene cut - byte for byte.
usage: enc_stress_gpu.py [quick]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_yuv import frame  # noqa: E402

# (width, height, frames, qp, seed, cut, intra period[, strong intra smoothing])
CASES = [(1920, 1080, 4, 32, 7, None, 2, 1), (832, 480, 6, 38, 11, None, 3, 1),
         (832, 480, 10, 30, 3, None, 100), (1280, 720, 8, 24, 5, None, 100), (1920, 1080, 12, 37, 9, None, 5), (1920, 1080, 10, 22, 2, None, 100),
         (416, 240, 30, 35, 8, 12, 100), (640, 360, 16, 28, 21, 7, 6), (1920, 1080, 10, 30, 4, 6, 100), (2560, 1440, 4, 33, 6, None, 100),
         (3840, 2160, 3, 27, 12, None, 100), (352, 288, 24, 20, 13, None, 4), (1024, 768, 9, 40, 17, 5, 100), (1920, 1080, 6, 45, 1, None, 100)]


def main():
    from x265_amd.encoder import encode_clip
    cases = CASES[:3] if len(sys.argv) > 1 and sys.argv[1] == "quick" else CASES
    bad = 0
    for case in cases:
        w, h, n, qp, seed, cut, ip = case[:7]
        sis = len(case) > 7 and case[7]
        rng = np.random.default_rng(seed)
        pics = [frame(w, h, t, rng, cut=cut) for t in range(n)]
        exe = os.path.join(ROOT, "oracle", "_ref", "x265_refsse_2160p" if h > 1088 else "x265_refsse")
        with tempfile.TemporaryDirectory() as d:
            yuv, out = os.path.join(d, "in.yuv"), os.path.join(d, "o.265")
            with open(yuv, "wb") as f:
                for p in pics:
                    f.write(b"".join(c.tobytes() for c in p))
            t0 = time.perf_counter()
            subprocess.run([exe, "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(n), "-q", str(qp), "-ip", str(ip), "-sceneChange",
                            "1" if cut is not None else "0", "-t", "16"] + (["-sis"] if sis else []), check=True, capture_output=True)
            t_ref = time.perf_counter() - t0
            ref = open(out, "rb").read()
        t0 = time.perf_counter()
        ours = encode_clip(pics, w, h, qp, intra_period=ip, scene_change=cut is not None, spin_us=200, strong_intra_smoothing=bool(sis))
        t_own = time.perf_counter() - t0
        same = ours == ref
        bad += not same
        print(f"{w}x{h} {n} pictures QP {qp} seed {seed} cut {cut} intra period {ip} sis {int(bool(sis))}: {'identical' if same else 'DIFFERENT'} "
              f"({len(ours)} / {len(ref)} bytes, md5 {hashlib.md5(ours).hexdigest()[:12]}; reference {t_ref:.2f} s, own {t_own:.2f} s)", flush=True)
    print("ALL IDENTICAL" if not bad else f"{bad} DIFFERENT")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
