#!/usr/bin/env python3
"""Run the encoder integration harness variants and compare bitstreams.
  enc_parity.py --write            (development container) record refsse md5s into tests/golden/encoder_md5.json
  enc_parity.py --add              the same for the configurations the file does not hold yet; the others stay as they are
  enc_parity.py --nosei            record the streams of the reference built without its CHECK_SEI option (x265_refsse_nosei:
                                   no picture-hash unit) into tests/golden/encoder_md5_nosei.json
  enc_parity.py VARIANT...         print md5/size/time per variant for every recorded config"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "encoder_md5.json")
# (width, height, frames, qp, scene cut at picture | None): with a cut the reference's scene-change detection is switched on
CONFIGS = [(416, 240, 8, 32, None), (416, 240, 3, 22, None), (416, 240, 8, 27, None), (416, 240, 8, 37, None), (416, 240, 8, 32, 6), (416, 240, 8, 32, 5),
           (1920, 1080, 4, 32, None), (1920, 1080, 3, 22, None), (1920, 1080, 3, 27, None), (1920, 1080, 3, 37, None),
           # the clips bench.py's encode legs are timed on: the published 64-picture 1080p configuration and the short 2160p clip
           (1920, 1080, 64, 32, None), (3840, 2160, 6, 32, None)]
SEED = {416: 1234, 1920: 7, 3840: 1234}     # generator seeds per picture width (tools/gen_yuv.py)


def run(variant, yuv, w, h, f, q, out, scene_change=0):
    # coded heights beyond the reference's MAX_HEIGHT: the build of its sources with that one constant raised (oracle/Makefile)
    exe = os.path.join(ROOT, "oracle", "_ref", f"x265_{variant}" + ("_2160p" if h > 1088 else ""))
    t = time.time()
    subprocess.run([exe, "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(f), "-q", str(q), "-ip", "100",
                    "-sceneChange", str(scene_change), "-t", "1"], check=True, capture_output=True)
    return hashlib.md5(open(out, "rb").read()).hexdigest(), os.path.getsize(out), time.time() - t


def write_nosei():
    gold = {}
    with tempfile.TemporaryDirectory() as d:
        for w, h, f, q in ((416, 240, 8, 32), (416, 240, 3, 22), (1920, 1080, 4, 32)):
            yuv = os.path.join(d, "in.yuv")
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(f), str(SEED[w]), yuv])
            m, n, t = run("refsse_nosei", yuv, w, h, f, q, os.path.join(d, "o.265"))
            print(f"{w}x{h}_f{f}_q{q} refsse_nosei {m} {n} B {t:.2f} s")
            gold[f"{w}x{h}_f{f}_q{q}"] = {"md5": m, "bytes": n, "input_md5": hashlib.md5(open(yuv, "rb").read()).hexdigest()}
    path = os.path.join(ROOT, "tests", "golden", "encoder_md5_nosei.json")
    json.dump(gold, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


def main():
    if "--nosei" in sys.argv:
        return write_nosei()
    add = "--add" in sys.argv
    write = "--write" in sys.argv or add
    variants = [a for a in sys.argv[1:] if not a.startswith("--")] or ["refsse", "oracle"]
    gold = {} if write and not add else json.load(open(GOLD))
    with tempfile.TemporaryDirectory() as d:
        for w, h, f, q, cut in CONFIGS:
            if add and f"{w}x{h}_f{f}_q{q}" + (f"_cut{cut}_scenechange" if cut is not None else "") in gold:
                continue
            yuv = os.path.join(d, "in.yuv")
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(f), str(SEED[w]), yuv]
                                  + ([str(cut)] if cut is not None else []))
            key = f"{w}x{h}_f{f}_q{q}" + (f"_cut{cut}_scenechange" if cut is not None else "")
            have = [v for v in variants if os.path.exists(os.path.join(ROOT, "oracle", "_ref", f"x265_{v}" + ("_2160p" if h > 1088 else "")))]
            res = {v: run(v, yuv, w, h, f, q, os.path.join(d, v + ".265"), 1 if cut is not None else 0) for v in have}
            for v, (m, n, t) in res.items():
                print(f"{key:20s} {v:8s} {m} {n:8d} B {t:7.2f} s")
            if write:
                assert "oracle" not in res or res["refsse"][:2] == res["oracle"][:2], "oracle primitives diverge from the reference's SSE primitives"
                gold[key] = {"md5": res["refsse"][0], "bytes": res["refsse"][1],
                             "input_md5": hashlib.md5(open(yuv, "rb").read()).hexdigest()}
            else:
                for v, (m, n, t) in res.items():
                    assert (m, n) == (gold[key]["md5"], gold[key]["bytes"]), f"{v} differs on {key}"
    if write:
        json.dump(gold, open(GOLD, "w"), indent=1, sort_keys=True)
        print("wrote", GOLD)


if __name__ == "__main__":
    main()
