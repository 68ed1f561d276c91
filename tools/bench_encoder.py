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


def clip(w, h, n, seed=None):
    rng = np.random.default_rng(seed if seed is not None else (7 if w == 1920 else 1234))
    return [frame(w, h, t, rng) for t in range(n)]


def recorded_stream(w, h, n, qp):
    """what tests/golden/encoder_md5.json holds for this clip (written by the reference-control-flow build, tools/enc_parity.py), or None"""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")) as f:
            return json.load(f).get(f"{w}x{h}_f{n}_q{qp}")
    except OSError:
        return None


def run(w=1920, h=1080, n=6, qp=32, lanes=2, spin_us=200, verbose=False, speculate=None, picture_hash=True):
    """one picture stream: the clip of the CPU baseline (tools/gen_yuv.py, seed 7 at 1080p), picture 0 intra; wall clock of the
    encode calls (pictures resident in host memory; the call uploads them)"""
    import hashlib
    from x265_amd.encoder import Encoder
    pics = clip(w, h, n)
    enc = Encoder(w, h, qp, lanes=lanes, spin_us=spin_us, speculate=speculate, picture_hash=picture_hash)
    try:
        warm = Encoder(w, h, qp)        # first call of a process: code objects, buffers - on an instance of its own, so that the
        warm.encode(*pics[0], True)     # timed instance starts from a new encoder's state, as the reference's run does
        warm.close()
        stream, per = b"", []
        t_all = time.perf_counter()
        for i, (y, u, v) in enumerate(pics):
            t0 = time.perf_counter()
            au = enc.encode(y, u, v, i == 0)
            dt = time.perf_counter() - t0
            stream += au
            per.append({"picture": i, "type": "I" if i == 0 else "P", "bytes": len(au), "ms": round(dt * 1e3, 2),
                        **{k: round(v, 2) if isinstance(v, float) else ({a: round(b, 1) for a, b in v.items()} if isinstance(v, dict) else v) for k, v in enc.timing().items()}})
            if verbose:
                print(json.dumps(per[-1]), flush=True)
        wall = time.perf_counter() - t_all
    finally:
        enc.close()
    p_ms = [p["ms"] for p in per[1:]]
    md5 = hashlib.md5(stream).hexdigest()
    want = recorded_stream(w, h, n, qp) if picture_hash else None
    return {"stream_matches_fixture": None if want is None else bool((want["md5"], want["bytes"]) == (md5, len(stream))),
            "what": f"{w}x{h} synthetic IPPP, {n} pictures, QP {qp}: this library's encoder (xmi_encoder_encode), one picture stream",
            "encode_fps": round(n / wall, 2), "pictures": n, "bytes": len(stream), "md5": md5,
            "i_picture_ms": per[0]["ms"], "i_picture_round_trips": per[0]["round_trips"], "i_picture_us_per_round_trip": per[0]["us_per_round_trip"], "p_picture_ms": round(float(np.mean(p_ms)), 2) if p_ms else None,
            "p_pictures_per_s": round(1e3 / float(np.mean(p_ms)), 2) if p_ms else None,
            "p_stage_ms": {k: round(float(np.mean([p[k] for p in per[1:]])), 2) for k in ("analysis_ms", "deblock_ms", "sao_ms", "write_ms")} if p_ms else None,
            "round_trips_per_p_picture": int(np.mean([p["round_trips"] for p in per[1:]])) if p_ms else None,
            "p_picture_us_per_round_trip": per[-1]["us_per_round_trip"] if p_ms else None}


def run_two_step(w=1920, h=1080, n=6, qp=32, lanes=2, spin_us=200):
    """the same stream with one picture in flight: picture t + 1 is handed in (xmi_encoder_submit) before the access unit of
    picture t is taken (xmi_encoder_collect), so the picture hash of t - a serial MD5 pass - runs beside the analysis of t + 1"""
    import hashlib
    from x265_amd.encoder import Encoder
    pics = clip(w, h, n)
    warm = Encoder(w, h, qp)
    warm.encode(*pics[0], True)
    warm.close()
    enc = Encoder(w, h, qp, lanes=lanes, spin_us=spin_us)
    try:
        stream = b""
        t0 = time.perf_counter()
        for i, (y, u, v) in enumerate(pics):
            enc.submit(y, u, v, i == 0)
            if i:
                stream += enc.collect()
        stream += enc.collect()
        wall = time.perf_counter() - t0
    finally:
        enc.close()
    want = recorded_stream(w, h, n, qp)
    md5 = hashlib.md5(stream).hexdigest()
    return {"stream_matches_fixture": None if want is None else bool((want["md5"], want["bytes"]) == (md5, len(stream))),
            "what": "the same clip through xmi_encoder_submit / xmi_encoder_collect, one picture in flight: an access unit is taken after the next picture was handed in",
            "encode_fps": round(n / wall, 2), "pictures": n, "bytes": len(stream), "md5": md5}


def run_segments(w=1920, h=1080, n=4, qp=32, segments=8, workers=None, spin_us=0, seed0=7, speculate=None):
    """`segments` independent GOP segments (each its own encoder, I picture first) encoded at the same time on one GPU, one
    host thread per segment (the C call releases the interpreter lock): whole-job pictures per second over all pictures,
    and over the P pictures alone (after a barrier behind the I pictures)"""
    import threading
    from x265_amd.encoder import Encoder
    clips = [clip(w, h, n, seed=seed0 + s) for s in range(segments)]
    encs = [Encoder(w, h, qp, spin_us=spin_us, workers=workers, speculate=speculate) for _ in range(segments)]
    errs, marks = [], {}
    gate = threading.Barrier(segments + 1)

    def work(s):
        try:
            encs[s].encode(*clips[s][0], True)
            gate.wait()
            gate.wait()
            for y, u, v in clips[s][1:]:
                encs[s].encode(y, u, v, False)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
            gate.abort()

    th = [threading.Thread(target=work, args=(s,)) for s in range(segments)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    try:
        gate.wait()
        marks["i_done"] = time.perf_counter()
        gate.wait()
    except threading.BrokenBarrierError:
        pass
    for t in th:
        t.join()
    t1 = time.perf_counter()
    for e in encs:
        e.close()
    if errs:
        return {"error": errs[0]}
    return {"segments": segments, "pictures_per_segment": n, "workers": workers, "spin_us": spin_us, "speculate": speculate, "encode_fps": round(segments * n / (t1 - t0), 2),
            "p_pictures_per_s": round(segments * (n - 1) / (t1 - marks["i_done"]), 2), "i_pictures_s": round(marks["i_done"] - t0, 3)}


def main():
    a = sys.argv[1:]
    if a and a[0] == "two_step":      # two_step W H FRAMES QP
        print(json.dumps(run_two_step(*(int(v) for v in a[1:5]))))
        return
    if a and a[0] == "segments":      # segments N PICTURES WORKERS SPIN_US [SPECULATE]
        n, pics, workers, spin = (int(v) for v in a[1:5])
        print(json.dumps(run_segments(1920, 1080, pics, 32, n, workers or None, spin, speculate=int(a[5]) if len(a) > 5 else None)))
        return
    w, h, n, qp = (int(v) for v in a[:4]) if len(a) >= 4 else (1920, 1080, 8, 32)
    lanes, spin = (int(a[4]), int(a[5])) if len(a) >= 6 else (2, 200)
    spec = int(a[6]) if len(a) >= 7 else None
    print(json.dumps(run(w, h, n, qp, lanes, spin, verbose=True, speculate=spec)))
    from x265_amd import capi
    L = capi.load()
    if hasattr(L, "xmi_debug_quad_ticks"):
        import ctypes as C
        qt = (C.c_uint64 * 6)()
        L.xmi_debug_quad_ticks(qt)
        if qt[5]:
            n = qt[5]
            print(json.dumps({"luma group of four 4x4 PUs, us": {"set-up": round(qt[0] / n / 100, 2), "neighbours x4": round(qt[1] / n / 100, 2),
                                                               "search x4": round(qt[2] / n / 100, 2), "prediction x4": round(qt[3] / n / 100, 2),
                                                               "chain x4": round(qt[4] / n / 100, 2)}}))
    if hasattr(L, "xmi_debug_round_classes"):    # measurement build: the calls of the P pictures by their largest CU
        import ctypes
        rc = (ctypes.c_ulonglong * 15)()
        L.xmi_debug_round_classes(rc)
        print(json.dumps({n: {"calls": int(rc[3 * i]), "us": round(rc[3 * i + 1] / max(1, rc[3 * i]) / 1e3, 1), "cus": round(rc[3 * i + 2] / max(1, rc[3 * i]), 1)}
                          for i, n in enumerate(("largest 8", "largest 16", "largest 32", "largest 64", "chroma alone"))}))
    if hasattr(L, "xmi_debug_merge64_hist"):
        import ctypes
        hh = (ctypes.c_ulonglong * 32)()
        L.xmi_debug_merge64_hist(hh)
        print(json.dumps({"64x64 merge path, workgroups by length (4 us bins)": [int(v) for v in hh[:16]], "by length of the merge selection (2 us bins)": [int(v) for v in hh[16:]]}))
    if hasattr(L, "xmi_debug_span"):
        import ctypes
        sp = (ctypes.c_ulonglong * 5)()
        L.xmi_debug_span(sp)
        n = max(1, sp[2])
        print(json.dumps({"rounds with 64x64 CUs, us on the GPU from the first workgroup's start": {"to the early flag": round(sp[0] / n / 100, 1), "to the full flag": round(sp[1] / n / 100, 1),
                          "longest early luma workgroup": round(sp[3] / n / 100, 1), "longest chroma workgroup": round(sp[4] / n / 100, 1), "rounds": int(sp[2])}}))
    if hasattr(L, "xmi_debug_intra_ticks"):      # measurement build (-DXMI_EXP_ROUND_TIMING): workgroup time per job type
        import ctypes
        tk = (ctypes.c_ulonglong * 18)()
        L.xmi_debug_intra_ticks(tk)
        names = ("pu4", "pu8", "pu16", "pu32", "chroma4", "chroma8", "chroma16", "luma_group", "chroma_group")
        print(json.dumps({n: {"us": round(tk[2 * i] / max(1, tk[2 * i + 1]) / 100.0, 2), "count": int(tk[2 * i + 1])} for i, n in enumerate(names)}))
    if "--segments" in a:
        print(json.dumps(run_segments(w, h, 4, qp, 8)))


if __name__ == "__main__":
    main()
