#!/usr/bin/env python3
"""GOP-segment encode over several ranks (TEST / MEASUREMENT INFRASTRUCTURE: it runs the integration-harness encoders
of oracle/_ref, i.e. the reference's control flow).

One process per rank (torch.distributed: "nccl" = RCCL on GPUs, "gloo" for CPU rehearsals). Rank r takes whole GOPs
[begin, end) of a synthetic clip (x265_amd.segments.partition_frames), encodes them with a fresh encoder state
- every segment starts with an I picture that re-emits VPS/SPS/PPS (src/encode.cpp:5789-5808) - and the only
exchange is the end-of-segment gather of per-picture statistics records and of the byte segments. Rank 0 joins the
segments in rank order into one valid stream and prints ONE JSON line. A sharded encode equals the reference run
separately on each segment (not one long run: the FAST_MD averages are never reset, src/encode.cpp:3556-3568), so the
parity fixture is one md5 per segment (tests/golden/segment_md5.json; --write-golden records it with x265_refsse).

usage: [python -m torch.distributed.run --nproc-per-node N ...] tools/encode_segments.py --variant gpu_me --frames 16 --gop 8 --check
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden", "segment_md5.json")


def clip_frames(w, h, n, seed):
    """The frames of tools/gen_yuv.py (noise drawn sequentially per frame) as a list of bytes objects."""
    import gen_yuv
    rng = np.random.default_rng(seed)
    return [b"".join(p.tobytes() for p in gen_yuv.frame(w, h, t, rng)) for t in range(n)]


def picture_sizes(stream):
    """Bytes of each coded picture: NAL units split at start codes; a picture = one VCL NAL (type < 32) plus the
    non-VCL units in front of it."""
    pos, starts = 0, []
    while True:
        i = stream.find(b"\x00\x00\x01", pos)
        if i < 0:
            break
        starts.append(i)
        pos = i + 3
    sizes, acc = [], starts[0] if starts else 0          # bytes in front of the first start code (a leading zero byte)
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(stream)
        acc += e - s
        if ((stream[s + 3] >> 1) & 0x3f) < 32:
            sizes.append(acc)
            acc = 0
    if sizes:
        sizes[-1] += acc                                   # suffix units after the last picture
    return sizes


def encode(variant, frames, w, h, qp, workdir):
    if variant == "own":      # the library's own encoder on this rank's GPU: nothing of the reference runs
        from x265_amd.encoder import encode_clip
        pics = []
        for fr in frames:
            a = np.frombuffer(fr, np.uint8)
            pics.append((a[:w * h].reshape(h, w), a[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), a[w * h * 5 // 4:].reshape(h // 2, w // 2)))
        t0 = time.perf_counter()
        data = encode_clip(pics, w, h, qp, intra_period=100000)
        return data, time.perf_counter() - t0
    exe = os.path.join(ROOT, "oracle", "_ref", f"x265_{variant}")
    yuv, out = os.path.join(workdir, "seg.yuv"), os.path.join(workdir, "seg.265")
    with open(yuv, "wb") as f:
        for fr in frames:
            f.write(fr)
    t0 = time.perf_counter()
    subprocess.run([exe, "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(len(frames)), "-q", str(qp), "-ip", "100000",
                    "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, timeout=3000)
    return open(out, "rb").read(), time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="refsse", help="own = the library's encoder; else oracle/_ref/x265_<variant>: refsse (CPU), gpu_me, gpu_planes, gpu_all")
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--gop", type=int, default=8)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--backend", default=os.environ.get("XMI_BENCH_BACKEND", "nccl"))
    ap.add_argument("--out")
    ap.add_argument("--check", action="store_true", help="compare every segment's md5 with tests/golden/segment_md5.json")
    ap.add_argument("--write-golden", action="store_true", help="single process: record the md5 of every GOP with x265_refsse (or the refsse* variant named)")
    ap.add_argument("--golden", default=GOLD, help="fixture file of --check / --write-golden")
    args = ap.parse_args()
    w, h = args.width, args.height
    key = f"{w}x{h}_q{args.qp}_seed{args.seed}_gop{args.gop}"

    if args.write_golden:
        frames = clip_frames(w, h, args.frames, args.seed)
        gold = json.load(open(args.golden)) if os.path.exists(args.golden) else {}
        writer = args.variant if args.variant.startswith("refsse") else "refsse"
        with tempfile.TemporaryDirectory() as d:
            for b in range(0, args.frames, args.gop):
                e = min(b + args.gop, args.frames)
                data, _ = encode(writer, frames[b:e], w, h, args.qp, d)
                gold.setdefault(key, {})[f"{b}-{e}"] = {"md5": hashlib.md5(data).hexdigest(), "bytes": len(data)}
        json.dump(gold, open(args.golden, "w"), indent=1, sort_keys=True)
        print("wrote", args.golden)
        return

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    comm_dev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            comm_dev = torch.device("cuda", local)
        else:
            dist.init_process_group(args.backend)
    from x265_amd import segments
    b, e = segments.partition_frames(args.frames, world, rank, args.gop)
    frames = clip_frames(w, h, e, args.seed)[b:e]
    # one encoder run per GOP of the span (fresh state each), on this rank's GPU
    if args.variant == "own":
        # the library runs inside this process: bind it to the rank's GPU (ranks of a gloo rehearsal share the devices there are)
        from x265_amd import capi
        L = capi.load()
        capi.check(L.xmi_init(local % max(1, L.xmi_device_count())))
    elif not args.variant.startswith("refsse") and world > 1 and os.environ.get("HIP_VISIBLE_DEVICES") is None:
        import torch
        # the in-situ encoders are child processes: each sees only its rank's GPU (set before the child starts, so it holds)
        os.environ["HIP_VISIBLE_DEVICES"] = str(local % max(1, torch.cuda.device_count()))
    stream, secs, md5s, recs = b"", 0.0, {}, []
    with tempfile.TemporaryDirectory() as d:
        for g0 in range(b, e, args.gop):
            g1 = min(g0 + args.gop, e)
            data, t = encode(args.variant, frames[g0 - b:g1 - b], w, h, args.qp, d)
            stream += data
            secs += t
            md5s[f"{g0}-{g1}"] = {"md5": hashlib.md5(data).hexdigest(), "bytes": len(data)}
            sizes = picture_sizes(data)
            assert len(sizes) == g1 - g0, f"{len(sizes)} pictures in the stream of frames {g0}..{g1}"
            fr = np.arange(g0, g1)
            lam = 0.578 * 2 ** ((args.qp - 12) / 3.0)
            recs.append(segments.make_records(fr, np.where(fr == g0, 2, 1), np.full(len(fr), args.qp), np.full(len(fr), lam),
                                              np.zeros(len(fr)), np.array(sizes) * 8))
    rec = np.concatenate(recs) if recs else np.zeros((0, len(segments.RC_FIELDS)))
    allrec = segments.gather_records(rec, device=comm_dev)
    parts = segments.gather_segments(stream, device=comm_dev)
    allmd5 = segments.gather_segments(json.dumps(md5s).encode(), device=comm_dev)
    alltime = segments.gather_segments(repr(secs).encode(), device=comm_dev)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    whole = b"".join(parts)
    assert allrec.shape[0] == args.frames and int(allrec[:, 5].sum()) == 8 * len(whole)
    got = {}
    for m in allmd5:
        got.update(json.loads(m.decode()))
    if args.out:
        open(args.out, "wb").write(whole)
    wall = max(float(t.decode()) for t in alltime)
    res = {"variant": args.variant, "ranks": world, "frames": args.frames, "gop": args.gop, "bytes": len(whole), "md5": hashlib.md5(whole).hexdigest(),
           "segments": got, "encode_seconds_max_rank": round(wall, 3), "fps": round(args.frames / wall, 3) if wall > 0 else None}
    if args.check:
        want = json.load(open(args.golden))[key]
        bad = {k: (v, want.get(k)) for k, v in got.items() if want.get(k) != v}
        res["parity"] = "identical" if not bad and len(got) == args.frames // args.gop else f"DIFFERS: {bad or 'segment count'}"
    print(json.dumps(res), flush=True)
    if args.check and res["parity"] != "identical":
        sys.exit(1)


if __name__ == "__main__":
    main()
