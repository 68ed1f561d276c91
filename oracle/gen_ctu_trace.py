#!/usr/bin/env python3
"""Writes tests/golden/ctu_trace.npz: P pictures of bit-identical encodes by the reference's encoder built with
oracle/ctu_trace_hook.cpp (make -C oracle ctu_trace): for each kept picture the inputs of the analysis loop - the padded
reference planes, the state carried in from earlier pictures - and everything the reference's CTU loop left behind: decisions
(info0 / info1 / vectors / vector differences / predictor indices), levels, the RD cost and estimation contexts of every CTU,
the state carried out, and the md5 of the unfiltered reconstruction planes. Of the source pictures only the margin the encoder
adds when it aligns the picture to CTU multiples (xFrameAligning, src/preProcess.cpp:33) is stored; tools/gen_yuv.py regenerates
the picture itself (md5 of the clip kept). Run in the development container; needs /root/reference."""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import ctu_trace as T  # noqa: E402

KEEP = [("416x240_q32", 416, 240, 8, 32, 1234, (1, 2, 7)), ("416x240_q22", 416, 240, 3, 22, 1234, (1,))]


def main():
    out = {}
    tags = []
    for name, w, h, frames, qp, seed, keep in KEEP:
        with tempfile.TemporaryDirectory() as d:
            pics, stream = T.record_trace(d, w, h, frames, qp, seed)
            clip_md5 = hashlib.md5(open(os.path.join(d, "in.yuv"), "rb").read()).hexdigest()
        for k in keep:
            pic, before = pics[k], pics[k - 1]
            assert pic["slice"] == 1
            t = f"{name}_p{k}"
            tags.append(t)
            out[f"{t}_meta"] = np.array([w, h, frames, qp, seed, k, pic["w"], pic["h"], pic["pad"], pic["rows"], pic["cols"]], np.int32)
            out[f"{t}_clip_md5"] = np.frombuffer(clip_md5.encode(), np.uint8)
            out[f"{t}_stream_md5"] = np.frombuffer(hashlib.md5(stream).hexdigest().encode(), np.uint8)
            for i, c in enumerate("yuv"):
                ww, hh = (w, h) if i == 0 else (w // 2, h // 2)
                out[f"{t}_src_right_{c}"] = pic["src"][i][:, ww:]
                out[f"{t}_src_below_{c}"] = pic["src"][i][hh:, :ww]
                out[f"{t}_ref_{c}"] = pic["ref"][i]
                out[f"{t}_rec_md5_{c}"] = np.frombuffer(hashlib.md5(np.ascontiguousarray(pic["rec"][i]).tobytes()).hexdigest().encode(), np.uint8)
            for key in ("sum", "count", "avg"):
                out[f"{t}_in_{key}"] = before[key]
                out[f"{t}_out_{key}"] = pic[key]
            out[f"{t}_in_frac"], out[f"{t}_out_frac"] = pic["frac_start"], pic["frac"]
            for key in ("info0", "info1", "mv", "mvd", "mvp_idx", "coef_y", "coef_u", "coef_v", "ctx_in", "ctx_out"):
                out[f"{t}_{key}"] = pic[key]
            out[f"{t}_rd_cost"] = pic["rd_cost"].astype(np.uint32)
    out["tags"] = np.array(tags)
    path = os.path.join(ROOT, "tests", "golden", "ctu_trace.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(tags), "pictures")


if __name__ == "__main__":
    main()
