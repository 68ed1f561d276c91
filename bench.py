#!/usr/bin/env python3
"""Benchmark of the ME SAD-surface hot path on MI355X.

A step = one P-frame: the full 85-PU SAD pyramid of every CTU over the [-R,R]^2 integer search window
(current frame vs. the padded previous frame), computed by libx265_amd.so from frames that are
already resident in HBM. Workload at N=1: 1920x1080 (coded 1920x1088, 510 CTUs), R=64, synthetic
IPPP video (the 1080p configuration the headline metric is quoted on). With --gpus N every rank runs the same number of steps on its
own GOP segment (weak scaling, no data-path collective); the only exchange is the final gather of
per-rank statistics.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Eight segment encoders in one process keep eight and more streams busy, and HIP's default of four hardware queues per process
# makes them take turns: measured with eight concurrent 1080p segments, their I pictures take 0.47 s with four queues and 0.38 s
# with eight (P pictures 115 -> 109 per second; 4-picture segments 47 -> 53 fps). Read by the HIP runtime when it starts,
# i.e. at the first GPU call below; nothing else in this file depends on it.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak; measured copy ceiling is ~6.3 TB/s
PAD = 80                   # the reference pads luma by MAX_CU_SIZE+16 (inc/config.h:52); grown to R+16 for R > 64


ENCODE_FRAMES = 64      # the encode legs (this library's encoder, and the reference's on the host cores) use the published
                        # configuration: 1920x1080, 64 pictures, IPPP, QP 32
UHD_FRAMES = 6          # pictures of the 3840x2160 encode legs
# host threads per encoder / analyser when eight of them share one GPU and the process's cores (xmi_encoder_workers): a
# handle's CTU rows are fibers of that many threads, and 8 x 2 of them fit the 16 cores a GPU's share of a host has
SEGMENT_WORKERS = 2


def host_core_slice(world, local):
    """-> (cores this rank may use, host threads per encoder when eight encoders share them)"""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1, SEGMENT_WORKERS
    limit = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)      # the GPU box exports its per-GPU core share this way
    if world > 1:
        per = max(1, len(cores) // world)
        mine = cores[local * per:(local + 1) * per] or cores[-per:]
        try:
            os.sched_setaffinity(0, mine)
            cores = mine
        except OSError:
            pass
    n = len(cores) if world > 1 or not limit else min(len(cores), limit)
    return n, max(1, n // 8)


def measured_traffic(w, h, rng):
    """HBM bytes per launch from the committed PMC passes of this workload (x265_amd/measured_traffic.json), else None.
    Counters cannot be collected from inside this process; tools/profile_round.sh refreshes the file."""
    try:
        with open(os.path.join(ROOT, "x265_amd", "measured_traffic.json")) as f:
            return json.load(f).get(f"{w}x{h}_R{rng}", {}).get("bytes_per_launch")
    except OSError:
        return None


def synth_frame(w, h, t, seed):
    """Synthetic 8-bit luma: moving sinusoid texture (3,2) px/frame + uniform noise in [-6,6]."""
    rng = np.random.default_rng(seed * 1000 + t)
    yy, xx = np.mgrid[0:h, 0:w]
    y = 128 + 60 * np.sin((xx + 3 * t) / 17.0) + 40 * np.cos((yy + 2 * t) / 11.0) + rng.integers(-6, 7, (h, w))
    return np.clip(y, 0, 255).astype(np.uint8)


def pad_frame(f, pad):
    return np.pad(f, pad, mode="edge")  # edge replication, as xFramePadding does (src/encode.cpp:2043)


def algorithmic_bytes(w, h, rng, pad):
    nctu = (w // 64) * (h // 64)
    nmv = (2 * rng + 1) ** 2
    return w * h + (w + 2 * pad) * (h + 2 * pad) + nctu * nmv * 180, nctu * nmv * 85


def cpu_baseline(cur, refp, rng, budget_s=12.0):
    """Time the CPU SAD path on a bounded sample of CTUs (rank 0 only). Returns a dict or None."""
    exe = os.path.join(ROOT, "oracle", "cpu_baseline")
    if not os.path.exists(exe):
        return None
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libx265ref_pixel.so")
    h, w = cur.shape
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(cur.tobytes())
        f.write(refp.tobytes())
        path = f.name
    try:
        def run(nctu):
            args = [exe, path, str(w), str(h), str(PAD), str(rng), str(nctu)]
            if os.path.exists(ref_lib):
                args.append(ref_lib)
            out = subprocess.run(args, capture_output=True, text=True, timeout=300).stdout.strip()
            kv = dict(x.split("=") for x in out.split())
            return kv["kind"], int(kv["cands"]), float(kv["seconds"])
        kind, cands, sec = run(2)                      # calibrate
        n = max(2, min((w // 64) * (h // 64), int(2 * budget_s / max(sec, 1e-6))))
        kind, cands, sec = run(n)
        out = {"value": round(cands / sec / 1e9, 5), "unit": "Gcand/s", "cores": 1, "kind": kind,
               "sample": f"{n} of {(w // 64) * (h // 64)} CTUs x {(2 * rng + 1) ** 2} MVs x 85 PUs, one xSad call per "
                         f"candidate, {sec:.1f} s single thread"}
        prims = cpu_primitives(ref_lib)
        if prims:
            out["primitives"] = prims
        enc = cpu_encoder_fps()
        if enc:
            out["encoder"] = enc
        uhd = cpu_encoder_fps(UHD_FRAMES, width=3840, height=2160, seed=1234)      # the clip of encode.uhd
        if uhd:
            out["encoder_2160p"] = uhd
        return out
    finally:
        os.unlink(path)


def cpu_primitives(ref_lib, seconds_per_case=0.25):
    """Per-block primitives on one host core (same thread budget as above): the reference's compiled functions
    where they exist, the restatement where the reference binds NASM. Mblk/s per primitive and size."""
    exe = os.path.join(ROOT, "oracle", "cpu_primitives")
    if not os.path.exists(exe):
        return None
    args = [exe, ref_lib if os.path.exists(ref_lib) else "", str(seconds_per_case)]
    lines = subprocess.run(args, capture_output=True, text=True, timeout=120).stdout.splitlines()
    rows = [json.loads(x) for x in lines if x.startswith("{")]
    return {r["cpu"]: {"Mblk_per_s": r["Mblk_per_s"], "GBps": r["GBps"], "kind": r["kind"]} for r in rows}


def cpu_encoder_fps(frames=ENCODE_FRAMES, qp=32, width=1920, height=1080, seed=7):
    """The reference's encoder (its own sources compiled into oracle/_ref/x265_refsse; the 7 NASM-only SATD/SSD entries
    come from the restatement) on a short 1080p IPPP clip, wall clock of the whole run: one thread, and the reference's
    own OpenMP row threads (x265_refsse_mt, -t up to its limit of 16) on the host cores this process may use."""
    big = height > 1088          # beyond the reference's MAX_HEIGHT: the build of its sources with that constant raised (oracle/Makefile)
    exe = os.path.join(ROOT, "oracle", "_ref", "x265_refsse_2160p" if big else "x265_refsse")
    exe_mt = os.path.join(ROOT, "oracle", "_ref", "x265_refsse_2160p_mt" if big else "x265_refsse_mt")
    gen = os.path.join(ROOT, "tools", "gen_yuv.py")
    if not (os.path.exists(exe) and os.path.exists(gen)):
        return None
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores) or cores)))
    with tempfile.TemporaryDirectory() as d:
        yuv, out = os.path.join(d, "in.yuv"), os.path.join(d, "out.265")
        subprocess.run([sys.executable, gen, str(width), str(height), str(frames), str(seed), yuv], check=True, timeout=300)

        def run(binary, threads):
            cmd = [binary, "-i", yuv, "-o", out, "-w", str(width), "-h", str(height), "-f", str(frames), "-q", str(qp), "-ip", "100",
                   "-sceneChange", "0", "-t", str(threads)]
            t0 = time.perf_counter()
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            sec = time.perf_counter() - t0
            if res.returncode != 0 or not os.path.exists(out):
                return None
            return round(frames / sec, 3), os.path.getsize(out)
        one = run(exe, 1)
        if not one:
            return None
        # The reference's encoder is timed as its command line runs: process start and reading the clip from the file included.
        # This library's encoder (the `encode` block) is timed on pictures already in host memory. To compare like with like, the
        # time this process needs to start a trivial child and to read the same file is measured here and taken off.
        t0 = time.perf_counter()
        subprocess.run(["/bin/true"], check=False)
        with open(yuv, "rb") as f:
            while f.read(1 << 24):
                pass
        overhead = time.perf_counter() - t0
        enc = {"fps": one[0], "frames": frames, "qp": qp, "threads": 1, "bytes": one[1],
               "what": f"{width}x{height} synthetic IPPP, reference control flow and C++/SSE4 primitives, file I/O included",
               "start_and_read_seconds": round(overhead, 3),
               "fps_without_start_and_read": round(frames / max(1e-9, frames / one[0] - overhead), 3)}
        threads = min(16, cores)
        if os.path.exists(exe_mt) and threads > 1:
            many = run(exe_mt, threads)
            if many and many[1] == one[1]:
                enc["fps_row_threads"] = many[0]
                enc["row_threads"] = threads
                enc["fps_row_threads_without_start_and_read"] = round(frames / max(1e-9, frames / many[0] - overhead), 3)
        return enc


def satd_rates(L, capi, torch, dev, stream, reps=10):
    """The other half of the headline metric: block SATD evaluations per second of the batch kernel (one evaluation = one
    N x N block pair; HM form = xCalcHADs*, x264 form = the xSatdN table entries), 512 Mi pixels per operand resident in
    HBM (1 GiB per launch, four times the last-level cache), HIP events on the launch stream. Algorithmic bytes per evaluation: 2 N^2 read + 4 written."""
    out = {}
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    for metric, name, sizes in ((capi.METRIC_SATD_HM, "hm", (8, 16, 32)), (capi.METRIC_SATD_X264, "x264", (4, 8, 16))):
        for n in sizes:
            cnt = (512 << 20) // (n * n)
            a = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device=dev, generator=g)
            b = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device=dev, generator=g)
            res = torch.empty(cnt, dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(reps + 2):
                if i == 2:
                    e0.record(stream)
                capi.check(L.xmi_block_metric_batch_dev(metric, n, cnt, a.data_ptr(), b.data_ptr(), res.data_ptr(), stream.cuda_stream))
            e1.record(stream)
            torch.cuda.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / reps
            gbs = cnt * (2 * n * n + 4) / sec / 1e9
            out[f"{name} {n}x{n}"] = {"Geval_per_s": round(cnt / sec / 1e9, 2), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
            del a, b, res
    return out


def analysis_rate(L, qp=32, pictures=4, segments=True, workers=SEGMENT_WORKERS):
    """The encode-side half of the headline metric with this library's code only: P pictures per second of
    xmi_analysis_p_picture (mode decision + reconstruction of every CTU of a 1080p picture: the reference's xEncodeCUs for
    inter pictures, host decision logic + GPU pixel work). Outside the timed region, rank 0. The reference's own encoder on
    the host cores is reported under cpu_baseline.encoder (fps = one thread, fps_row_threads = its OpenMP row threads)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import bench_analysis
        out = bench_analysis.run(1920, 1080, qp, pictures, lib=L)
        # GOP segments are independent (each starts at an I picture), so several are analysed at the same time on one GPU,
        # one handle and one host thread group each; the aggregate is what a segment-parallel encode sees
        # With several analysers on one GPU the pixel work counts, not the round trips: how far requests reach ahead
        # (xmi_analysis_speculate) is the caller's choice, so both settings are measured and the better one is quoted.
        if segments:
            by_level = {}
            for level in (1, 2):
                seg = bench_analysis.run_segments(1920, 1080, qp, 3, 8, lib=L, speculate=level, workers=workers)
                by_level[level] = seg.get("analysis_fps", seg.get("error"))
            ok = {k: v for k, v in by_level.items() if isinstance(v, (int, float))}
            out["analysis_fps_8_segments"] = max(ok.values()) if ok else by_level[2]
            out["analysis_fps_8_segments_by_speculate"] = {str(k): v for k, v in by_level.items()}
        return out
    except Exception as e:      # noqa: BLE001 - reported, the headline measurement above stands on its own
        return {"error": repr(e)}


def encode_rate(L, qp=32, segments=True, workers=SEGMENT_WORKERS):
    """The first half of the headline metric with this library's code only: encode fps of xmi_encoder_encode (analysis loop,
    in-loop filters, SAO, entropy coding; I and P pictures; the stream is the reference's, tests/test_encoder_gpu.py) on the
    clip the reference's encoder is timed on under cpu_baseline.encoder. Outside the timed region, rank 0."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import bench_encoder
        out = bench_encoder.run(1920, 1080, ENCODE_FRAMES, qp)
        out["two_step"] = bench_encoder.run_two_step(1920, 1080, ENCODE_FRAMES, qp)
        # the reference's build without its CHECK_SEI option (no picture-hash unit; src/CMakeLists.txt:37): another stream, pinned
        # by tests/golden/encoder_md5_nosei.json at other clip lengths
        nh = bench_encoder.run(1920, 1080, ENCODE_FRAMES, qp, picture_hash=False)
        out["without_picture_hash"] = {k: nh[k] for k in ("encode_fps", "bytes", "p_picture_ms", "p_stage_ms")}
        if segments:
            out["segments_8"] = bench_encoder.run_segments(1920, 1080, 4, qp, 8, workers=workers)
            # the 3840x2160 configuration (coded 3840x2176), a short clip: to be read beside cpu_baseline.encoder_2160p
            uhd = bench_encoder.run(3840, 2160, UHD_FRAMES, qp)
            out["uhd"] = {k: uhd[k] for k in ("stream_matches_fixture", "what", "encode_fps", "pictures", "bytes", "md5", "i_picture_ms", "p_picture_ms", "p_stage_ms",
                                              "round_trips_per_p_picture") if k in uhd}
            two = bench_encoder.run_two_step(3840, 2160, UHD_FRAMES, qp)
            out["uhd"]["two_step"] = {k: two[k] for k in ("stream_matches_fixture", "encode_fps")}
        return out
    except Exception as e:      # noqa: BLE001 - reported, the headline measurement above stands on its own
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed back-to-back launches after the warm-up steps until this much time has passed: the first ~30 ms of a "
                         "burst run at boost clocks and then through a throttling overshoot (380 -> 530 -> 360 us per launch)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--range", type=int, default=64)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames kept resident per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="diagnostic runs: leave out the satd and analysis blocks")
    ap.add_argument("--no-analysis", action="store_true", help="leave out the analysis block")
    ap.add_argument("--uhd-segment-frames", type=int, default=0,
                    help="with --gpus N > 1: also encode one 3840x2160 GOP segment of this many pictures per rank, all ranks at once "
                         "(the sharded configuration; 16 is its published shape, 0 leaves the leg out)")
    ap.add_argument("--no-segments", action="store_true",
                    help="leave out the 8-segment leg of the analysis block (rocprofv3's kernel trace crashes inside hipLaunchKernel when "
                         "eight host threads launch at once)")
    args = ap.parse_args()
    global PAD
    PAD = max(PAD, args.range + 16)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: start the ranks (before anything here touches the GPU) and leave with
        # their exit status. The driver's torch.distributed.run command line sets WORLD_SIZE and never comes here.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Host cores: the encode legs are host-bound (CTU-row fibers, entropy coding), so with several ranks on one host every rank
    # takes its own slice of the cores this job may use - before anything here touches the GPU or starts a thread - and sizes
    # the host-thread groups of its eight concurrent encoders to it. One rank keeps whatever it was given.
    cores_per_rank, segment_workers = host_core_slice(world, local)

    import torch
    import torch.distributed as dist
    from x265_amd import capi

    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    # XMI_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("XMI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    L = capi.load()
    capi.check(L.xmi_init(local))

    w = args.width
    h = (args.height + 63) // 64 * 64          # frames are padded to CTU multiples (src/x265.cpp:157-168)
    w = (w + 63) // 64 * 64
    R = args.range
    from x265_amd import layout
    nctu = (w // 64) * (h // 64)
    nmv1 = 2 * R + 1

    # each rank owns its own GOP segment of synthetic video (seed per rank)
    frames = [synth_frame(w, h, t, seed=7 + rank) for t in range(args.frames)]
    d_cur = [torch.from_numpy(f).to(dev) for f in frames]
    d_ref = [torch.from_numpy(pad_frame(f, PAD)).to(dev) for f in frames]
    ref_stride = w + 2 * PAD

    s8 = torch.empty(layout.plane_elems(nctu, R, 0), dtype=torch.int16, device=dev)
    s16 = torch.empty(layout.plane_elems(nctu, R, 1), dtype=torch.int16, device=dev)
    s32 = torch.empty(layout.plane_elems(nctu, R, 2), dtype=torch.int32, device=dev)
    s64 = torch.empty(layout.plane_elems(nctu, R, 3), dtype=torch.int32, device=dev)
    geom = capi.SadGeom(w, h, w, ref_stride, R)
    stream = torch.cuda.current_stream(dev)

    def step(i):
        t = 1 + i % (args.frames - 1)          # P-frame t is searched in frame t-1
        ref0 = d_ref[t - 1].data_ptr() + PAD * ref_stride + PAD
        capi.check(L.xmi_sad_surface_dev(C.byref(geom), d_cur[t].data_ptr(), ref0, s8.data_ptr(), s16.data_ptr(),
                                         s32.data_ptr(), s64.data_ptr(), 0, nctu, stream.cuda_stream))

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t_settle, n_settle = time.perf_counter(), 0
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for i in range(32):
            step(n_settle + i)
        n_settle += 32
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(args.steps):
        step(i)
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # average duration of the dominant kernel (one launch per step), HIP events on the launch stream
    kern_ms_raw = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    kern_ms = sorted(kern_ms_raw)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    # end-of-segment exchange (outside the timed region), the only collective of the path (RCCL when world > 1): one
    # statistics record per processed frame, gathered on every rank. No bitstream exists in this loop, so the column
    # that carries coded bits in an encode (tools/encode_segments.py) carries the frame's matching cost here: the sum
    # over its CTUs of the smallest 64x64 SAD in the window, read back from the surface the step produced. A second
    # set of records carries the kernel time of every step.
    from x265_amd import segments
    pitch64, nmain = layout.row_pitch(R, 3), nctu * nmv1 * layout.row_pitch(R, 3)
    cost = {}
    for t in range(1, args.frames):
        step(t - 1)
        torch.cuda.synchronize()
        main = s64[:nmain].view(nctu, nmv1, pitch64)[:, :, :2 * R].reshape(nctu, -1).min(dim=1).values
        edge = s64[nmain:nmain + nctu * nmv1].view(nctu, nmv1).min(dim=1).values
        cost[t] = int(torch.minimum(main, edge).sum().item())
    fr = np.arange(args.steps) + rank * args.steps
    qp = 32
    rec = segments.make_records(fr, np.ones(args.steps), np.full(args.steps, qp), np.full(args.steps, 0.578 * 2 ** ((qp - 12) / 3.0)),
                                np.zeros(args.steps), np.array([cost[1 + i % (args.frames - 1)] for i in range(args.steps)], np.float64))
    allrec = segments.gather_records(rec, device=comm_dev)
    assert allrec.shape[0] == args.steps * world
    trec = segments.make_records(fr, np.ones(args.steps), np.zeros(args.steps), np.zeros(args.steps), np.zeros(args.steps), np.array(kern_ms_raw) * 1e3)
    alltime = segments.gather_records(trec, device=comm_dev)
    # per-rank view of the same records: average kernel time and the device each rank ran on
    per_rank = []
    for r in range(world):
        mine = alltime[(alltime[:, 0] >= r * args.steps) & (alltime[:, 0] < (r + 1) * args.steps)]
        per_rank.append(round(float(mine[:, 5].mean()) * 1e-3, 4))
    devs = [None] * world
    if world > 1:
        dist.all_gather_object(devs, f"cuda:{local}" + ("" if backend == "nccl" else " (shared, gloo rehearsal)"))
    else:
        devs = [f"cuda:{local}"]
    # encode leg of a multi-GPU run: every rank encodes its own GOP segments with the library's encoder at the same time
    # (8 concurrent segments per GPU, as in the single-GPU encode block); rank 0 reports the sum. No exchange inside.
    enc_ranks = uhd_ranks = None
    if world > 1 and not args.headline_only and not args.no_analysis:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_encoder
        dist.barrier()
        try:
            mine = bench_encoder.run_segments(1920, 1080, 4, 32, 8, workers=segment_workers)
        except Exception as e:      # noqa: BLE001
            mine = {"error": repr(e)}
        enc_ranks = [None] * world
        dist.all_gather_object(enc_ranks, mine)
        # the sharded 3840x2160 configuration: one GOP segment per rank and GPU (--uhd-segment-frames pictures each, the
        # published shape is 16), all ranks at the same time, every rank with all the cores of its slice
        uhd_ranks = None
        if args.uhd_segment_frames > 0:
            dist.barrier()
            try:
                one = bench_encoder.run_segments(3840, 2160, args.uhd_segment_frames, 32, 1, workers=None, seed0=9 + rank)
            except Exception as e:      # noqa: BLE001
                one = {"error": repr(e)}
            uhd_ranks = [None] * world
            dist.all_gather_object(uhd_ranks, one)
    if world > 1:              # all collectives are done: leave the group together, before rank 0's CPU-side legs
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        abytes, cands = algorithmic_bytes(w, h, R, PAD)
        value = cands * args.steps * world / elapsed / 1e9
        achieved = abytes / (kern_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "ME SAD candidate evaluations per second (85-PU pyramid over the full integer window)",
            "value": round(value, 3), "unit": "Gcand/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.width}x{args.height} 8-bit luma (coded {w}x{h}, {nctu} CTUs), IPPP P-frame SAD pyramid, "
                                   f"search range {R} ({nmv1}x{nmv1} integer MVs), {args.frames} resident frames per GPU",
                       "cands_per_step": cands, "frames_per_s": round(args.steps * world / elapsed, 2),
                       "parallelism": f"gop-segments x{world}", "cores_per_rank": cores_per_rank, "workers": segment_workers,
                       "gpu_max_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       # what becomes of the surface: the library's encoder replays the reference's sparse diamond walk (about ten of
                       # the 16 641 vectors per PU) and computes those SADs itself; the surface serves exhaustive searches
                       "consumed_by_encoder": False, "rank_devices": devs, "rank_kernel_ms_avg": per_rank,
                       "gathered_records": int(allrec.shape[0]), "matching_cost_per_frame_mean": round(float(allrec[:, 5].mean()), 1)},
            # bound: the figure is bytes against the HBM peak. What holds the kernel below it, by the counters and clock probes under
            # profiles/: store path (5.4 TB/s ceiling for streaming stores) and SAD issue rate are within 10 % of each other, and with
            # both running the chip lowers its clock from 2.39 to 2.04 GHz (either half alone keeps 2.39).
            "roofline": {"bound": "hbm", "limiter": "stores + SAD issue under a power-lowered clock (2.04 of 2.39 GHz)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(w, h, R),
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms_avg": round(kern_avg_ms, 4),
                         "kernel_ms_min": round(kern_ms[0], 4)},
        }
        # the same kernel against its arithmetic bound: every (pixel, MV) pair is one absolute difference, and the SAD
        # instructions of gfx950 retire 64 of them per clock and SIMD (v_qsad_pk_u16_u8: 16 per lane every 16 clocks,
        # tools/microbench/qsad_cycles.hip); 1024 SIMDs at the 2.4 GHz peak clock. Streaming stores of this shape top out at
        # 5.4 TB/s (tools/microbench/store_pitch4.hip), so the two bounds lie within 25 % of each other.
        absdiff = float(w) * h * nmv1 * nmv1
        valu_peak = 256 * 4 * 64 * 2.4
        out["roofline_valu_sad"] = {"bound": "valu_sad", "achieved": round(absdiff / (kern_avg_ms * 1e-3) / 1e9, 1), "peak": valu_peak,
                                    "unit": "G abs-diff/s", "frac": round(absdiff / (kern_avg_ms * 1e-3) / 1e9 / valu_peak, 4), "clock_ghz": 2.4,
                                    "store_ceiling_GBps": 5400.0}
        out["config"]["settle_ms"] = args.settle_ms
        if not args.headline_only:
            out["satd"] = satd_rates(L, capi, torch, dev, stream)
            if not args.no_analysis:
                out["analysis"] = analysis_rate(L, segments=not args.no_segments, workers=segment_workers)
                out["encode"] = encode_rate(L, segments=not args.no_segments, workers=segment_workers)
                if enc_ranks:
                    good = [r for r in enc_ranks if r and "error" not in r]
                    failed = {str(i): (r or {}).get("error", "no result") for i, r in enumerate(enc_ranks) if not r or "error" in r}
                    if failed:      # a sum over fewer ranks than ran is not a whole-job figure: say which ranks failed and why
                        out["encode"]["all_ranks_8_segments_error"] = failed
                    out["encode"]["all_ranks_8_segments"] = {"p_pictures_per_s": round(sum(r["p_pictures_per_s"] for r in good), 2),
                                                             "encode_fps": round(sum(r["encode_fps"] for r in good), 2), "ranks": len(good),
                                                             "per_rank_p_pictures_per_s": [r.get("p_pictures_per_s") if r else None for r in enc_ranks]}
                if uhd_ranks:
                    good = [r for r in uhd_ranks if r and "error" not in r]
                    out["encode"]["all_ranks_2160p_one_segment_each"] = {
                        "what": f"3840x2160, one GOP segment of {args.uhd_segment_frames} pictures per rank, all ranks at the same time",
                        "encode_fps": round(sum(r["encode_fps"] for r in good), 2), "ranks": len(good),
                        "per_rank_encode_fps": [r.get("encode_fps") if r else None for r in uhd_ranks],
                        "errors": {str(i): (r or {}).get("error", "no result") for i, r in enumerate(uhd_ranks) if not r or "error" in r} or None}
        if not args.no_cpu_baseline:
            t = 1
            out["cpu_baseline"] = cpu_baseline(frames[t], pad_frame(frames[t - 1], PAD), R)
            # vs_baseline stays null: the reference publishes no figure for this metric. Ratios against the CPU figures measured
            # beside it on this box (reported baselines, not targets):
            cb = out["cpu_baseline"] or {}
            ratios = {}
            if cb.get("value"):
                ratios["sad_candidates_vs_one_core"] = round(value / world / cb["value"], 1)
            enc, ref = out.get("encode") or {}, cb.get("encoder") or {}
            if enc.get("encode_fps") and ref.get("fps_without_start_and_read"):
                ratios["encode_fps_1080p_vs_reference_one_thread"] = round(enc["encode_fps"] / ref["fps_without_start_and_read"], 2)
            if enc.get("encode_fps") and ref.get("fps_row_threads_without_start_and_read"):
                ratios["encode_fps_1080p_vs_reference_row_threads"] = round(enc["encode_fps"] / ref["fps_row_threads_without_start_and_read"], 2)
            if ratios:
                out["vs_cpu_baseline"] = ratios
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
