#!/usr/bin/env python3
"""Synthetic 8-bit 4:2:0 planar YUV (the generator of the benchmark, plus chroma): usage gen_yuv.py W H FRAMES SEED out.yuv [CUT]
With CUT, frames CUT.. show a different scene (for the reference's scene-change detection, src/preProcess.cpp:104)."""
import sys

import numpy as np


def frame(w, h, t, rng, cut=None):
    yy, xx = np.mgrid[0:h, 0:w]
    if cut is not None and t >= cut:      # another scene: different texture, brightness and motion
        y = np.clip(90 + 70 * np.cos((xx - 2 * t) / 7.0) * np.sin((yy + t) / 5.0) + rng.integers(-6, 7, (h, w)), 0, 255)
    else:
        y = np.clip(128 + 60 * np.sin((xx + 3 * t) / 17.0) + 40 * np.cos((yy + 2 * t) / 11.0) + rng.integers(-6, 7, (h, w)), 0, 255)
    y2, x2 = np.mgrid[0:h:2, 0:w:2]
    u = np.clip(128 + 30 * np.sin((x2 + 3 * t) / 23.0), 0, 255)
    v = np.clip(128 + 30 * np.cos((y2 + 2 * t) / 19.0), 0, 255)
    return y.astype(np.uint8), u.astype(np.uint8), v.astype(np.uint8)


if __name__ == "__main__":
    w, h, n, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    cut = int(sys.argv[6]) if len(sys.argv) > 6 else None
    rng = np.random.default_rng(seed)
    with open(out, "wb") as f:
        for t in range(n):
            for p in frame(w, h, t, rng, cut):
                f.write(p.tobytes())
