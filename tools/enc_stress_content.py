#!/usr/bin/env python3
"""GPU box: the library's encoder against the reference-control-flow build on kinds of content the synthetic benchmark clip
is not: still pictures, fast and diagonal motion, pure noise, hard edges at the ends of the sample range, a moving block over a
gradient, flat pictures. Byte for byte. usage: enc_stress_content.py"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def chroma(w, h, t, rng, kind):
    y2, x2 = np.mgrid[0:h // 2, 0:w // 2]
    if kind in ("noise", "extremes"):
        return rng.integers(0, 256, (h // 2, w // 2)).astype(np.uint8), rng.integers(0, 256, (h // 2, w // 2)).astype(np.uint8)
    u = np.clip(128 + 40 * np.sin((x2 + 5 * t) / 9.0) + rng.integers(-2, 3, (h // 2, w // 2)), 0, 255)
    v = np.clip(128 + 40 * np.cos((y2 - 3 * t) / 13.0) + rng.integers(-2, 3, (h // 2, w // 2)), 0, 255)
    return u.astype(np.uint8), v.astype(np.uint8)


def luma(w, h, t, rng, kind, still):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "still":
        return still
    if kind == "fast":          # 23 samples to the right and 11 down per picture: beyond the clip box near the borders
        return np.clip(128 + 70 * np.sin((xx - 23 * t) / 13.0) * np.cos((yy - 11 * t) / 9.0) + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == "extremes":      # 0 / 255 checkerboards of 12 samples that move by 5: the largest residuals there are
        return (((xx + 5 * t) // 12 + (yy + 3 * t) // 12) % 2 * 255).astype(np.uint8)
    if kind == "block":         # a textured block crossing a gradient: occlusion, different vectors side by side
        y = (xx * 255 // w).astype(np.int32) + rng.integers(-1, 2, (h, w))
        bx, by = (20 + 9 * t) % (w - 80), (30 + 4 * t) % (h - 80)
        y[by:by + 72, bx:bx + 72] = 60 + 100 * (((xx[by:by + 72, bx:bx + 72] - bx) // 6 + (yy[by:by + 72, bx:bx + 72] - by) // 6) % 2)
        return np.clip(y, 0, 255).astype(np.uint8)
    if kind == "grain":         # film grain of +-25 over a moving texture: every TU has coefficients, few CUs skip
        return np.clip(128 + 60 * np.sin((xx + 3 * t) / 11.0) * np.cos((yy - 2 * t) / 8.0) + rng.integers(-25, 26, (h, w)), 0, 255).astype(np.uint8)
    if kind == "flat":
        return np.full((h, w), 77 + t, np.uint8)
    raise ValueError(kind)


# (kind, width, height, pictures, qp, intra period)
# pure noise at a low QP needs more than the reference's 16384 CABAC commands of one CTU (MAX_CMDBUF_SIZE, inc/x265.h:416) and
# its build with assertions stops there (src/encode.cpp:5645); those cases only show that this encoder goes on
CASES = [("grain", 416, 240, 5, 30, 100), ("grain", 640, 360, 4, 20, 100), ("noise", 416, 240, 3, 48, 100), ("still", 416, 240, 6, 30, 100), ("fast", 640, 360, 8, 28, 100), ("noise", 416, 240, 4, 35, 100), ("noise", 192, 128, 3, 8, 2),
         ("extremes", 416, 240, 5, 22, 100), ("extremes", 256, 192, 4, 40, 3), ("block", 832, 480, 10, 26, 100), ("flat", 320, 192, 5, 32, 100),
         ("fast", 1920, 1080, 5, 33, 100), ("block", 1920, 1080, 6, 38, 100)]


def main():
    from x265_amd.encoder import encode_clip
    bad = 0
    for kind, w, h, n, qp, ip in CASES:
        rng = np.random.default_rng(len(kind) * 1000 + w + qp)
        still = np.clip(128 + 90 * np.sin(np.mgrid[0:h, 0:w][1] / 7.0) + rng.integers(-8, 9, (h, w)), 0, 255).astype(np.uint8)
        pics = []
        for t in range(n):
            y = luma(w, h, t, rng, kind, still)
            u, v = chroma(w, h, 0 if kind == "still" else t, np.random.default_rng(5) if kind == "still" else rng, kind)
            pics.append((y, u, v))
        with tempfile.TemporaryDirectory() as d:
            yuv, out = os.path.join(d, "in.yuv"), os.path.join(d, "o.265")
            with open(yuv, "wb") as f:
                for p in pics:
                    f.write(b"".join(c.tobytes() for c in p))
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "x265_refsse"), "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(n), "-q", str(qp),
                                "-ip", str(ip), "-sceneChange", "0", "-t", "16"], capture_output=True)
            ref = open(out, "rb").read() if r.returncode == 0 else None
        ours = encode_clip(pics, w, h, qp, intra_period=ip, spin_us=200, lanes=2)
        if ref is None:     # the reference itself stops on this input (its command buffer of one CTU is full): nothing to compare with
            print(f"{kind} {w}x{h} {n} pictures QP {qp} intra period {ip}: the reference build ends with status {r.returncode} "
                  f"({r.stderr.decode(errors='replace').strip().splitlines()[-1:] or ''}); this encoder wrote {len(ours)} bytes", flush=True)
            continue
        same = ours == ref
        bad += not same
        print(f"{kind} {w}x{h} {n} pictures QP {qp} intra period {ip}: {'identical' if same else 'DIFFERENT'} ({len(ours)} / {len(ref)} bytes, "
              f"md5 {hashlib.md5(ours).hexdigest()[:12]})", flush=True)
    print("ALL IDENTICAL" if not bad else f"{bad} DIFFERENT")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
