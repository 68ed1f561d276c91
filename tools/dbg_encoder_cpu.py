#!/usr/bin/env python3
"""Development aid: encode a synthetic clip with the host encoder on the CPU checker (oracle/libx265_analysis_check.so) and
with the reference-control-flow build (oracle/_ref/x265_refsse), and compare the streams access unit by access unit.
usage: dbg_encoder_cpu.py W H FRAMES QP [SEED]"""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_yuv import frame  # noqa: E402


def align_frame(y, u, v, W, H):
    """The picture the reference's command line hands its encoder when the size is no multiple of 64 (xFrameAligning,
    src/preProcess.cpp:33): columns are padded with the row's last sample; the added rows are copies of W bytes of the input
    buffer from the start of the plane's last row on - the row itself and, past its end, whatever follows it in the
    Y, U, V buffer (zeros after V)."""
    h, w = y.shape
    flat = np.concatenate([y.ravel(), u.ravel(), v.ravel(), np.zeros(W, np.uint8)])
    out, start = [], 0
    for p, s in ((y, 1), (u, 2), (v, 2)):
        ph, pw, PW, PH = p.shape[0], p.shape[1], W // s, H // s
        a = np.zeros((PH, PW), np.uint8)
        a[:ph, :pw] = p
        a[:ph, pw:] = p[:, -1:]
        if PH > ph:
            last = start + (ph - 1) * pw
            a[ph:, :] = flat[last:last + PW]
        out.append(np.ascontiguousarray(a))
        start += ph * pw
    return out


def split_aus(b):
    """access units of this encoder's streams: every slice NAL (type 1 or 19) starts one, parameter sets belong to the next"""
    starts = []
    i = 0
    while True:
        i = b.find(b"\x00\x00\x01", i)
        if i < 0:
            break
        starts.append(i)
        i += 3
    aus, cur = [], 0
    for k, s in enumerate(starts):
        t = (b[s + 3] >> 1) & 0x3f
        if t in (39, 40) and k + 1 < len(starts):      # SEI ends an access unit
            nxt = starts[k + 1]
            while nxt > 0 and b[nxt - 1] == 0:
                nxt -= 1
            aus.append(b[cur:nxt])
            cur = nxt
    aus.append(b[cur:])
    return [a for a in aus if a]


def main():
    w, h, n, qp = (int(a) for a in sys.argv[1:5])
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1234
    W, H = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libx265_analysis_check.so"))
    lib.orc_encoder_create.restype = ctypes.c_void_p
    lib.orc_encoder_create.argtypes = [ctypes.c_int] * 4
    lib.orc_encoder_encode.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    lib.orc_encoder_configure.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    enc = lib.orc_encoder_create(W, H, qp, 0)
    rng = np.random.default_rng(seed)
    out = np.zeros(W * H * 3 + 8192, np.uint8)
    ours = []
    with tempfile.TemporaryDirectory() as d:
        yuv, ref = os.path.join(d, "in.yuv"), os.path.join(d, "ref.265")
        with open(yuv, "wb") as f:
            for t in range(n):
                y, u, v = frame(w, h, t, rng)
                f.write(y.tobytes() + u.tobytes() + v.tobytes())
                ya, ua, va = align_frame(y, u, v, W, H)
                t0 = time.time()
                size = lib.orc_encoder_encode(enc, ya.ctypes.data, ua.ctypes.data, va.ctypes.data, 2 if t == 0 else 1, out.ctypes.data, out.size)
                assert size > 0, size
                ours.append(bytes(out[:size]))
                print(f"picture {t}: {size} bytes, {time.time() - t0:.2f} s", flush=True)
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "x265_refsse"), "-i", yuv, "-o", ref, "-w", str(w), "-h", str(h), "-f", str(n), "-q", str(qp),
                        "-ip", "100", "-sceneChange", "0", "-t", "1"], check=True, capture_output=True)
        refb = open(ref, "rb").read()
    aus = split_aus(refb)
    print(len(aus), "reference access units", [len(a) for a in aus])
    ok = True
    for t, (a, b) in enumerate(zip(ours, aus)):
        if a == b:
            print(f"picture {t}: identical ({len(a)} bytes)")
            continue
        ok = False
        k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
        print(f"picture {t}: DIFFERS at byte {k} (ours {len(a)} bytes, reference {len(b)})")
        print("  ours", a[max(0, k - 8):k + 16].hex())
        print("  ref ", b[max(0, k - 8):k + 16].hex())
        break
    print("ALL IDENTICAL" if ok and len(ours) == len(aus) else "MISMATCH")
    return ok


def compare_trace(w, h, n, qp, seed, which):
    """decisions of picture `which` against the CTU trace of the reference's encoder"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctu_trace
    W, H = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    with tempfile.TemporaryDirectory() as d:
        pics, _ = ctu_trace.record_trace(d, w, h, n, qp, seed)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libx265_analysis_check.so"))
    lib.orc_encoder_create.restype = ctypes.c_void_p
    lib.orc_encoder_create.argtypes = [ctypes.c_int] * 4
    lib.orc_encoder_encode.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    lib.orc_encoder_analysis.argtypes = [ctypes.c_void_p] * 12
    enc = lib.orc_encoder_create(W, H, qp, 0)
    out = np.zeros(W * H * 3 + 8192, np.uint8)
    for t in range(which + 1):
        pic = pics[t]
        ya, ua, va = (np.ascontiguousarray(p) for p in pic["src"])
        size = lib.orc_encoder_encode(enc, ya.ctypes.data, ua.ctypes.data, va.ctypes.data, 2 if t == 0 else 1, out.ctypes.data, out.size)
        assert size > 0, size
    pic = pics[which]
    rows, cols = pic["rows"], pic["cols"]
    g = {"info0": np.zeros((H // 4, W // 4), np.uint16), "info1": np.zeros((H // 4, W // 4), np.uint8), "mode": np.zeros((H // 4, W // 4), np.uint8),
         "coef_y": np.zeros((H, W), np.int16), "coef_u": np.zeros((H // 2, W // 2), np.int16), "coef_v": np.zeros((H // 2, W // 2), np.int16),
         "rec_y": np.zeros((H, W), np.uint8), "rec_u": np.zeros((H // 2, W // 2), np.uint8), "rec_v": np.zeros((H // 2, W // 2), np.uint8),
         "ctx_out": np.zeros((rows, cols, 256), np.uint8), "frac": np.zeros(rows, np.uint64)}
    lib.orc_encoder_analysis(enc, *[a.ctypes.data for a in g.values()])
    for k, ref in (("rec_y", pic["rec"][0]), ("rec_u", pic["rec"][1]), ("rec_v", pic["rec"][2]), ("coef_y", pic["coef_y"]), ("coef_u", pic["coef_u"]),
                   ("coef_v", pic["coef_v"]), ("info1", pic["info1"]), ("info0", pic["info0"]), ("ctx_out", pic["ctx_out"]), ("frac", pic["frac"])):
        a = g[k]
        if k == "info0" and pic["slice"] == 2:      # only the first entry of a PU is meaningful in I pictures: compare where ours is non-zero... all
            pass
        bad = np.argwhere(a != ref)
        print(f"{k:8s}", "identical" if not len(bad) else f"{len(bad)} differences, first at {tuple(bad[0])}: ours {a[tuple(bad[0])]} reference {ref[tuple(bad[0])]}")



if __name__ == "__main__":
    if "--trace" in sys.argv:
        i = sys.argv.index("--trace")
        which = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
        w, h, n, qp = (int(a) for a in sys.argv[1:5])
        compare_trace(w, h, n, qp, int(sys.argv[5]) if len(sys.argv) > 5 else 1234, which)
    else:
        main()
