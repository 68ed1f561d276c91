#!/usr/bin/env python3
"""P pictures per second of xmi_analysis_p_picture (decision + reconstruction of every CTU, this library's code only) on
synthetic video: picture t is analysed against the padded source of picture t-1 as its reference (no loop filters in
between: the analysis work per picture is what is measured, not a conforming encode). Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synth(w, h, t, seed):
    rng = np.random.default_rng(seed * 1000 + t)
    yy, xx = np.mgrid[0:h, 0:w]
    y = np.clip(128 + 60 * np.sin((xx + 3 * t) / 17.0) + 40 * np.cos((yy + 2 * t) / 11.0) + rng.integers(-6, 7, (h, w)), 0, 255).astype(np.uint8)
    y2, x2 = np.mgrid[0:h // 2, 0:w // 2]
    u = np.clip(128 + 30 * np.sin((2 * x2 + 3 * t) / 23.0), 0, 255).astype(np.uint8)
    v = np.clip(128 + 30 * np.cos((2 * y2 + 2 * t) / 19.0), 0, 255).astype(np.uint8)
    return y, u, v


def run_segments(width=1920, height=1080, qp=32, pictures=4, segments=4, lanes=None, spin_us=0, lib=None, speculate=None, workers=None):
    """`segments` independent GOP segments analysed at the same time on one GPU (one handle, one host thread each; the C
    call releases the interpreter lock): aggregate P pictures per second."""
    import threading
    from x265_amd import capi
    from tests import ctu_trace as T
    L = lib or capi.load()
    capi.check(L.xmi_init(0))
    w, h = (width + 63) // 64 * 64, (height + 63) // 64 * 64
    clips = [[synth(w, h, t, 7 + s) for t in range(pictures + 1)] for s in range(segments)]
    refs = [[tuple(np.pad(p, 80 if i == 0 else 40, mode="edge") for i, p in enumerate(f)) for f in clip] for clip in clips]
    runners = [T.Runner(L, "xmi_analysis_", w, h, lanes, spin_us, speculate=speculate, workers=workers) for _ in range(segments)]
    errs = []

    def work(s, lo, hi):
        try:
            for t in range(lo, hi):
                pic = {"w": w, "h": h, "qp": qp, "pad": 80, "rows": h // 64, "cols": w // 64, "src": clips[s][t], "ref": refs[s][t - 1]}
                runners[s].run(pic)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    def phase(lo, hi):
        th = [threading.Thread(target=work, args=(s, lo, hi)) for s in range(segments)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    phase(1, 2)                                   # warm-up picture per segment
    sec = phase(2, pictures + 1)
    for r in runners:
        r.close()
    if errs:
        return {"error": errs[0]}
    return {"segments": segments, "lanes": lanes, "workers": workers, "speculate": 2 if speculate is None else speculate, "analysis_fps": round(segments * (pictures - 1) / sec, 3), "qp": qp,
            "what": f"{segments} GOP segments of {width}x{height} analysed concurrently on one GPU, aggregate P pictures per second"}


def run(width=1920, height=1080, qp=32, pictures=6, seed=7, lanes=2, spin_us=200, lib=None, speculate=None):
    from x265_amd import capi
    from tests import ctu_trace as T
    L = lib or capi.load()
    capi.check(L.xmi_init(0))
    w, h = (width + 63) // 64 * 64, (height + 63) // 64 * 64
    frames = [synth(w, h, t, seed) for t in range(pictures + 1)]
    r = T.Runner(L, "xmi_analysis_", w, h, lanes, spin_us, speculate=speculate)
    times, rounds, cus = [], [], []
    try:
        for t in range(1, pictures + 1):
            pic = {"w": w, "h": h, "qp": qp, "pad": 80, "rows": h // 64, "cols": w // 64, "src": frames[t],
                   "ref": tuple(np.pad(p, 80 if i == 0 else 40, mode="edge") for i, p in enumerate(frames[t - 1]))}
            t0 = time.perf_counter()
            out = r.run(pic)
            times.append(time.perf_counter() - t0)
            rounds.append(int(out["counters"][0]))
            cus.append(int(out["counters"][1]))
        tm = (C.c_uint64 * 7)()
        L.xmi_analysis_timing.argtypes = [C.c_void_p, C.c_void_p]
        L.xmi_analysis_timing(r.handle, tm)
        if hasattr(L, "xmi_debug_round_ticks"):      # measurement build (-DXMI_EXP_ROUND_TIMING)
            tk = (C.c_uint64 * 32)()
            L.xmi_debug_round_ticks(tk)
            for si, n in enumerate((8, 16, 32, 64)):
                for path, name in enumerate(("merge path", "search path")):
                    m, sr, ch, cnt = (tk[(si * 2 + path) * 4 + k] for k in range(4))
                    if cnt:
                        print(f"{n}x{n} {name}: {cnt} workgroups, us each: merge selection {m / cnt / 100:.2f}, search {sr / cnt / 100:.2f}, "
                              f"chain {ch / cnt / 100:.2f}", file=sys.stderr)
        if hasattr(L, "xmi_debug_search_ticks"):
            stt = (C.c_uint64 * 10)()
            L.xmi_debug_search_ticks(stt)
            for i, n in enumerate((32, 64)):
                cnt = stt[i * 5 + 4]
                if cnt:
                    print(f"workgroup search {n}x{n}: {cnt} searches, us each: first stage {stt[i * 5] / cnt / 100:.2f}, diamond walk {stt[i * 5 + 1] / cnt / 100:.2f}, "
                          f"sub-pel rounds {stt[i * 5 + 2] / cnt / 100:.2f}; integer SADs {stt[i * 5 + 3] / cnt:.1f}", file=sys.stderr)
        if hasattr(L, "xmi_debug_clock_ticks"):
            ct = (C.c_uint64 * 2)()
            L.xmi_debug_clock_ticks(ct)
            if ct[1]:
                print(f"shader clock inside the round kernel: {ct[0] / ct[1] * 100:.0f} MHz (s_memtime ticks per 100 MHz tick)", file=sys.stderr)
        if hasattr(L, "xmi_debug_chain_ticks"):
            ck = (C.c_uint64 * 24)()
            L.xmi_debug_chain_ticks(ck)
            for si, n in enumerate((4, 8, 16, 32)):
                cnt = ck[si * 6 + 5]
                if cnt:
                    print(f"chain {n}x{n}: {cnt} units, us each: set-up {ck[si * 6] / cnt / 100:.2f}, forward {ck[si * 6 + 1] / cnt / 100:.2f}, "
                          f"quantiser {ck[si * 6 + 2] / cnt / 100:.2f}, inverse {ck[si * 6 + 3] / cnt / 100:.2f}, output {ck[si * 6 + 4] / cnt / 100:.2f}",
                          file=sys.stderr)
    finally:
        r.close()
    steady = times[1:] if len(times) > 1 else times
    return {"analysis_fps": round(len(steady) / sum(steady), 3), "ms_per_picture": [round(x * 1e3, 2) for x in times], "round_trips": rounds,
            "cu_requests": cus, "us_per_round_trip": round(sum(steady) / sum(rounds[-len(steady):]) * 1e6, 1), "qp": qp, "lanes": lanes, "spin_us": spin_us,
            "us_per_round_trip_parts": {"job_tables": round(tm[0] / max(1, tm[4]) / 1e3, 1), "launch_call": round(tm[1] / max(1, tm[4]) / 1e3, 1),
                                        "gpu_wait": round(tm[2] / max(1, tm[4]) / 1e3, 1), "read_back": round(tm[3] / max(1, tm[4]) / 1e3, 1)},
            "row_thread_us_per_request": round(tm[5] / max(1, tm[6]) / 1e3, 1), "row_requests_last_picture": int(tm[6]),
            "what": f"{width}x{height} synthetic P pictures, decision + reconstruction of every CTU; includes the host copies of the call "
                    "(planes up, result arrays down)"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--pictures", type=int, default=6)
    ap.add_argument("--segments", type=int, default=1)
    ap.add_argument("--lanes", type=int, default=None)
    ap.add_argument("--spin-us", type=int, default=None)
    ap.add_argument("--speculate", type=int, default=None)
    a = ap.parse_args()
    if a.segments > 1:
        print(json.dumps(run_segments(a.width, a.height, a.qp, a.pictures, a.segments, a.lanes, a.spin_us or 0, speculate=a.speculate)))
    else:
        print(json.dumps(run(a.width, a.height, a.qp, a.pictures, lanes=a.lanes or 2, spin_us=200 if a.spin_us is None else a.spin_us, speculate=a.speculate)))
