#!/usr/bin/env python3
"""Synthetic 8-bit 4:2:0 planar YUV (the generator of the benchmark, plus chroma): usage gen_yuv.py W H FRAMES SEED out.yuv"""
import sys

import numpy as np


def frame(w, h, t, rng):
    yy, xx = np.mgrid[0:h, 0:w]
    y = np.clip(128 + 60 * np.sin((xx + 3 * t) / 17.0) + 40 * np.cos((yy + 2 * t) / 11.0) + rng.integers(-6, 7, (h, w)), 0, 255)
    y2, x2 = np.mgrid[0:h:2, 0:w:2]
    u = np.clip(128 + 30 * np.sin((x2 + 3 * t) / 23.0), 0, 255)
    v = np.clip(128 + 30 * np.cos((y2 + 2 * t) / 19.0), 0, 255)
    return y.astype(np.uint8), u.astype(np.uint8), v.astype(np.uint8)


if __name__ == "__main__":
    w, h, n, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rng = np.random.default_rng(seed)
    with open(out, "wb") as f:
        for t in range(n):
            for p in frame(w, h, t, rng):
                f.write(p.tobytes())
