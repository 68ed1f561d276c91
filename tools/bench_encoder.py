#!/usr/bin/env python3
"""Encode a synthetic clip with the library's encoder on the GPU and print per-picture stage times and the encode rate.
usage: bench_encoder.py [W H FRAMES QP [LANES SPIN_US]]   (default 1920 1080 8 32)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_yuv import frame  # noqa: E402


def main():
    a = sys.argv[1:]
    w, h, n, qp = (int(v) for v in a[:4]) if len(a) >= 4 else (1920, 1080, 8, 32)
    lanes, spin = (int(a[4]), int(a[5])) if len(a) >= 6 else (1, 200)
    from x265_amd.encoder import Encoder
    rng = np.random.default_rng(7 if w == 1920 else 1234)
    pics = [frame(w, h, t, rng) for t in range(n)]
    enc = Encoder(w, h, qp, lanes=lanes, spin_us=spin)
    total, per = 0, []
    t_all = time.perf_counter()
    for i, (y, u, v) in enumerate(pics):
        t0 = time.perf_counter()
        au = enc.encode(y, u, v, i == 0)
        dt = time.perf_counter() - t0
        total += len(au)
        per.append({"picture": i, "type": "I" if i == 0 else "P", "bytes": len(au), "ms": round(dt * 1e3, 2), **{k: round(v, 2) if isinstance(v, float) else v for k, v in enc.timing().items()}})
        print(json.dumps(per[-1]), flush=True)
    wall = time.perf_counter() - t_all
    p_ms = [p["ms"] for p in per[1:]]
    print(json.dumps({"pictures": n, "bytes": total, "fps": round(n / wall, 2), "i_picture_ms": per[0]["ms"], "p_picture_ms_mean": round(float(np.mean(p_ms)), 2) if p_ms else None,
                      "p_fps": round(1e3 / float(np.mean(p_ms)), 2) if p_ms else None}))
    enc.close()


if __name__ == "__main__":
    main()
