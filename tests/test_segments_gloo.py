"""CPU: the N>1 host path (GOP-segment partition + record/segment gather) with world_size 2 over gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from tests.prereq import needs_ref

from x265_amd import segments

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_all_frames_once():
    for n, world, gop in ((64, 8, None), (128, 8, 16), (63, 4, 16), (5, 8, None), (100, 3, 8)):
        spans = [segments.partition_frames(n, world, r, gop) for r in range(world)]
        seen = []
        for b, e in spans:
            assert 0 <= b <= e <= n
            if gop:
                assert b % gop == 0 or b == n
            seen += list(range(b, e))
        assert seen == list(range(n))


def test_single_process_gather_is_identity():
    rec = segments.make_records([2, 0, 1], [1, 2, 1], [32, 32, 32], [10.0, 9.0, 10.0], [5, 50, 5], [100, 900, 120])
    out = segments.gather_records(rec)
    assert list(out[:, 0]) == [0, 1, 2]
    assert segments.gather_segments(b"abc") == [b"abc"]


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.environ["XROOT"])
    import numpy as np, torch.distributed as dist
    from x265_amd import segments
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    b, e = segments.partition_frames(21, world, rank, gop=8)
    fr = np.arange(b, e)
    rec = segments.make_records(fr, np.where(fr % 8 == 0, 2, 1), np.full(len(fr), 32), fr * 0.5, fr % 8 == 0, 1000 + fr)
    allrec = segments.gather_records(rec)
    assert allrec.shape == (21, 6) and list(allrec[:, 0]) == list(range(21)), allrec[:, 0]
    assert np.allclose(allrec[:, 5], 1000 + np.arange(21))
    segs = segments.gather_segments(bytes([rank + 1]) * (3 + 4 * rank))
    assert segs == [bytes([1]) * 3, bytes([2]) * 7], segs
    empty = segments.gather_segments(b"" if rank == 0 else b"xy")
    assert empty == [b"", b"xy"]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, XROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("ok") == 2


REFSSE = os.path.join(ROOT, "oracle", "_ref", "x265_refsse")


UHD = ["--width", "3840", "--height", "2160", "--frames", "4", "--gop", "2", "--seed", "11",
       "--golden", os.path.join(ROOT, "tests", "golden", "segment_md5_2160p.json")]


def run_segment_encode(variant, nproc, extra_env=None, clip=("--frames", "16", "--gop", "8")):
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1", XMI_BENCH_BACKEND="gloo", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "encode_segments.py"), "--variant", variant, *clip, "--check"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [x for x in res.stdout.splitlines() if x.startswith("{")][-1]
    return json.loads(line)


@needs_ref("x265_refsse", how="make -C oracle enc")
def test_gop_segments_on_two_ranks_join_into_the_per_segment_reference_streams():
    """Two ranks encode one GOP each with a fresh encoder (the reference's control flow, CPU primitives); the gathered
    byte segments are the streams recorded per segment, and the joined stream is the same as one rank produces."""
    two = run_segment_encode("refsse", 2)
    assert two["parity"] == "identical" and two["ranks"] == 2 and set(two["segments"]) == {"0-8", "8-16"}
    one = run_segment_encode("refsse", 1)
    assert one["md5"] == two["md5"] and one["bytes"] == two["bytes"]


@needs_ref("x265_refsse_2160p", how="make -C oracle enc_2160p")
def test_2160p_segments_on_two_ranks():
    """3840x2160 (coded 3840x2176, 2 040 CTUs), the first two of the eight 2-frame segments recorded in
    tests/golden/segment_md5_2160p.json, one per rank. The encoder is the reference's sources built against a header copy
    whose MAX_HEIGHT reads 2176 (oracle/Makefile, enc_2160p): the unmodified constant refuses this height."""
    two = run_segment_encode("refsse_2160p", 2, clip=UHD)
    assert two["parity"] == "identical" and two["ranks"] == 2 and set(two["segments"]) == {"0-2", "2-4"}
