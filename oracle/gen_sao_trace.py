#!/usr/bin/env python3
"""Record the SAO statistics of a real reference encode (TEST INFRASTRUCTURE; needs /root/reference).

`make -C oracle trace` builds oracle/_ref/x265_sao_trace (oracle/sao_trace_hook.cpp interposed on calcSaoStatsCu). This
script encodes the 416x240 IPPP clip at QP 37 with it, checks the bitstream against tests/golden/encoder_md5.json and
stores, for the I picture and one P picture, the deblocked reconstruction, the source picture, the count / offset
tables of every CTU and component, and (processSaoUnitAll) the SAO parameters of every CTU with the planes after the
offsets were applied, in tests/golden/sao_trace.npz."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KEEP = (0, 5)


def main():
    exe = os.path.join(HERE, "_ref", "x265_sao_trace")
    if not os.path.exists(exe):
        sys.exit("build it first: make -C oracle trace")
    qp = 37
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f8_q{qp}"]
    with tempfile.TemporaryDirectory() as d:
        yuv, out, tr = (os.path.join(d, n) for n in ("in.yuv", "out.265", "trace.bin"))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), "416", "240", "8", "1234", yuv])
        subprocess.run([exe, "-i", yuv, "-o", out, "-w", "416", "-h", "240", "-f", "8", "-q", str(qp), "-ip", "100",
                        "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, env=dict(os.environ, XMI_SAO_TRACE=tr))
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == gold["md5"], "traced encode differs from the recorded bitstream"
        raw = open(tr, "rb").read()
    pos, out, poc, pocs = 0, {}, None, []
    stats = {}
    while pos < len(raw):
        tag, a, b, c = (int(v) for v in np.frombuffer(raw, np.int32, 4, pos))
        pos += 16
        if tag == 1:
            w, h, poc = a, b, c
            planes = []
            for k in range(6):
                n = w * h if k % 3 == 0 else w * h // 4
                planes.append(np.frombuffer(raw, np.uint8, n, pos).reshape((h, w) if k % 3 == 0 else (h // 2, w // 2)))
                pos += n
            if poc in KEEP:
                pocs.append(poc)
                for k, name in enumerate(("rec_y", "rec_u", "rec_v", "org_y", "org_u", "org_v")):
                    out[f"p{poc}_{name}"] = planes[k]
                stats[poc] = []
        elif tag == 3:
            comp, nctu = a, b
            prm = np.frombuffer(raw, np.int32, nctu * 8, pos).reshape(nctu, 8)
            pos += nctu * 32
            shape = (h, w) if comp == 0 else (h // 2, w // 2)
            after = np.frombuffer(raw, np.uint8, shape[0] * shape[1], pos).reshape(shape)
            pos += shape[0] * shape[1]
            if c in KEEP:
                out[f"p{c}_param_{'yuv'[comp]}"] = prm
                out[f"p{c}_sao_{'yuv'[comp]}"] = after
                print(f"poc {c} component {comp}: types {sorted(set(prm[:, 0].tolist()))}, merge-left {int(prm[:, 6].sum())}, merge-up {int(prm[:, 7].sum())}, "
                      f"samples changed {int((after != out[f'p{c}_rec_' + 'yuv'[comp]]).sum())}")
        else:
            cnt = np.frombuffer(raw, np.int64, 165, pos).reshape(5, 33)
            off = np.frombuffer(raw, np.int64, 165, pos + 1320).reshape(5, 33)
            pos += 2640
            if poc in KEEP:
                stats[poc].append((a, b, c, cnt, off))
    for poc, lst in stats.items():
        out[f"p{poc}_ctu"] = np.array([r[:3] for r in lst], np.int32)          # ctu x, ctu y, component
        out[f"p{poc}_count"] = np.stack([r[3] for r in lst]).astype(np.int32)
        out[f"p{poc}_offset"] = np.stack([r[4] for r in lst]).astype(np.int32)
        print(f"poc {poc}: {len(lst)} (CTU, component) records, samples counted {int(out[f'p{poc}_count'].sum())}")
    out["pocs"] = np.array(pocs, np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sao_trace.npz"), **out)


if __name__ == "__main__":
    main()
