"""GPU: the library's own encoder (xmi_encoder_*: analysis loop, in-loop filters, SAO decision, CABAC writer) writes the
streams recorded in tests/golden/encoder_md5.json - the output of the reference's control flow on CPU primitives
(oracle/_ref/x265_refsse, see tools/enc_parity.py) - byte for byte, for I and P pictures, at 416x240 and 1920x1080 and every
QP of the published sweep. Nothing of the reference runs here: only the library, through its C ABI."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_yuv import frame  # noqa: E402

from tests.prereq import needs_ref  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))
SEED = {416: 1234, 1920: 7, 3840: 1234}


def clip(w, h, n):
    rng = np.random.default_rng(SEED[w])
    return [frame(w, h, t, rng) for t in range(n)]


@pytest.mark.parametrize("key", [k for k in sorted(GOLD) if "scenechange" not in k])
def test_stream_equals_the_recorded_stream(key):
    from x265_amd.encoder import encode_clip
    size, f, q = key.split("_")
    w, h = (int(v) for v in size.split("x"))
    frames, qp = int(f[1:]), int(q[1:])
    pics = clip(w, h, frames)
    raw = b"".join(p.tobytes() for pic in pics for p in pic)
    assert hashlib.md5(raw).hexdigest() == GOLD[key]["input_md5"], "the generator no longer makes the recorded input"
    stream = encode_clip(pics, w, h, qp)
    assert len(stream) == GOLD[key]["bytes"], f"{len(stream)} bytes, recorded {GOLD[key]['bytes']}"
    assert hashlib.md5(stream).hexdigest() == GOLD[key]["md5"]


@pytest.mark.parametrize("cut", [6, 5])
def test_scene_change_detection_of_the_command_line(cut):
    """A clip whose scene changes at picture `cut`, encoded with the reference's scene-change detection on (its default): the
    detection (src/preProcess.cpp:104, on the library's SAD) turns picture 6 into an I picture; the stream is the recorded one.
    Cut at 6: the first picture at which the detection runs (more than four P pictures behind the I picture). Cut at 5: the
    detection does not run at picture 5 and still fires at picture 6, because it compares with the reconstruction of the
    picture before the last one (the slot it reads holds the last picture's reference by then)."""
    from x265_amd.encoder import encode_clip
    key = f"416x240_f8_q32_cut{cut}_scenechange"
    rng = np.random.default_rng(1234)
    pics = [frame(416, 240, t, rng, cut=cut) for t in range(8)]
    raw = b"".join(p.tobytes() for pic in pics for p in pic)
    assert hashlib.md5(raw).hexdigest() == GOLD[key]["input_md5"], "the generator no longer makes the recorded input"
    stream = encode_clip(pics, 416, 240, 32, scene_change=True)
    assert len(stream) == GOLD[key]["bytes"], f"{len(stream)} bytes, recorded {GOLD[key]['bytes']}"
    assert hashlib.md5(stream).hexdigest() == GOLD[key]["md5"]
    assert stream.count(b"\x00\x00\x00\x01\x40\x01") == 2, "two I pictures, each with its parameter sets"


@needs_ref("x265_refsse", how="make -C oracle enc")
def test_strong_intra_smoothing_and_short_intra_period_against_the_reference_build(tmp_path):
    """-sis (bStrongIntraSmoothing: the SPS flag and the bilinear smoothing of 32x32 neighbours) with an I picture every third
    picture, a configuration no recorded stream covers: the reference-control-flow build runs beside the library on the box."""
    import subprocess
    from x265_amd.encoder import encode_clip
    w, h, n, qp = 832, 480, 6, 36
    rng = np.random.default_rng(11)
    pics = [frame(w, h, t, rng) for t in range(n)]
    yuv, out = str(tmp_path / "in.yuv"), str(tmp_path / "ref.265")
    with open(yuv, "wb") as f:
        for p in pics:
            f.write(b"".join(c.tobytes() for c in p))
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "x265_refsse"), "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(n), "-q", str(qp),
                    "-ip", "3", "-sceneChange", "0", "-t", "1", "-sis"], check=True, capture_output=True, timeout=300)
    ref = open(out, "rb").read()
    ours = encode_clip(pics, w, h, qp, intra_period=3, strong_intra_smoothing=True)
    assert len(ours) == len(ref) and ours == ref
    assert encode_clip(pics, w, h, qp, intra_period=3, strong_intra_smoothing=False) != ours, "the switch changes the stream (SPS flag)"


def test_2160p_segments_equal_the_recorded_streams():
    """The 3840x2160 configuration through the library's own encoder: two GOP segments (I + P, a new encoder each) of the clip
    recorded in tests/golden/segment_md5_2160p.json - written by the reference's sources built with MAX_HEIGHT 2176 on CPU
    primitives (oracle/Makefile, enc_2160p) - byte for byte. Coded size 3840x2176: 34 CTU rows of 60."""
    from x265_amd.encoder import encode_clip
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "segment_md5_2160p.json")))["3840x2160_q32_seed11_gop2"]
    rng = np.random.default_rng(11)
    pics = [frame(3840, 2160, t, rng) for t in range(4)]
    for b in (0, 2):
        stream = encode_clip(pics[b:b + 2], 3840, 2160, 32)
        want = gold[f"{b}-{b + 2}"]
        assert len(stream) == want["bytes"], f"segment {b}: {len(stream)} bytes, recorded {want['bytes']}"
        assert hashlib.md5(stream).hexdigest() == want["md5"], f"segment {b}"


def test_2160p_eight_segments_at_the_same_time_on_one_gpu():
    """What one rank of the sharded 3840x2160 configuration does, eight times over on this box's GPU: all eight GOP segments
    of tests/golden/segment_md5_2160p.json as eight encoder handles that encode at the same time (a host thread and two
    worker threads each, sleeping waiters). Every stream is the recorded one."""
    import threading
    from x265_amd.encoder import encode_clip
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "segment_md5_2160p.json")))["3840x2160_q32_seed11_gop2"]
    assert len(gold) == 8
    rng = np.random.default_rng(11)
    pics = [frame(3840, 2160, t, rng) for t in range(16)]
    got, errs = {}, []

    def work(b):
        try:
            got[b] = encode_clip(pics[b:b + 2], 3840, 2160, 32, workers=2, spin_us=0)
        except Exception as e:      # noqa: BLE001
            errs.append((b, repr(e)))

    th = [threading.Thread(target=work, args=(b,)) for b in range(0, 16, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for b in range(0, 16, 2):
        want = gold[f"{b}-{b + 2}"]
        assert len(got[b]) == want["bytes"], f"segment {b}: {len(got[b])} bytes, recorded {want['bytes']}"
        assert hashlib.md5(got[b]).hexdigest() == want["md5"], f"segment {b}"


def test_speculation_depth_does_not_change_the_stream():
    from x265_amd.encoder import encode_clip
    pics = clip(416, 240, 4)
    a = encode_clip(pics, 416, 240, 27)
    for level in (0, 1):
        assert encode_clip(pics, 416, 240, 27, speculate=level) == a, level


def test_gop_segments_on_two_ranks_with_the_library_encoder():
    """tools/encode_segments.py, two ranks (gloo rendezvous, both on this box's GPU), each encoding its GOP with the library's
    own encoder: every gathered segment is the stream recorded for it (tests/golden/segment_md5.json), at 416x240 and for two
    2-picture segments of the 3840x2160 fixture."""
    from tests.test_segments_gloo import UHD, run_segment_encode
    res = run_segment_encode("own", 2)
    assert res["parity"] == "identical" and res["ranks"] == 2 and set(res["segments"]) == {"0-8", "8-16"}
    res = run_segment_encode("own", 2, clip=UHD)
    assert res["parity"] == "identical" and res["ranks"] == 2 and set(res["segments"]) == {"0-2", "2-4"}


def test_worker_threads_do_not_change_the_stream():
    """the CTU rows as fibers of one or two host threads instead of one thread per row"""
    from x265_amd.encoder import encode_clip
    pics = clip(416, 240, 3)
    a = encode_clip(pics, 416, 240, 32)
    for workers, spin in ((1, 0), (2, 40)):
        assert encode_clip(pics, 416, 240, 32, workers=workers, spin_us=spin) == a, workers


def test_lanes_and_polling_do_not_change_the_stream():
    from x265_amd.encoder import encode_clip
    pics = clip(416, 240, 3)
    a = encode_clip(pics, 416, 240, 32)
    b = encode_clip(pics, 416, 240, 32, lanes=2, spin_us=50)
    assert a == b


def test_early_answers_at_1080p_write_the_recorded_stream():
    """With polling (spin_us > 0) a round of a P picture hands its answers out when the merge paths of the asking CUs and the
    chroma jobs are through, and the rows wait for the other parts when they get to them: two and three coordinators on the
    recorded 1920x1080 clip, where most rounds carry the long 64x64 searches that nobody then waits for."""
    from x265_amd.encoder import encode_clip
    key = next(k for k in sorted(GOLD) if k.startswith("1920x1080") and k.endswith("q32"))
    frames = int(key.split("_")[1][1:])
    pics = clip(1920, 1080, frames)
    for lanes, spin in ((2, 200), (3, 30)):
        stream = encode_clip(pics, 1920, 1080, 32, lanes=lanes, spin_us=spin)
        assert len(stream) == GOLD[key]["bytes"] and hashlib.md5(stream).hexdigest() == GOLD[key]["md5"], (lanes, spin)


def test_two_step_encode_writes_the_recorded_stream():
    """xmi_encoder_submit / xmi_encoder_collect with one picture in flight (the hash of picture t beside the analysis of
    t + 1): the recorded 64-picture 1920x1080 stream and a 416x240 one; the limits of the queue are errors."""
    from x265_amd.capi import XmiError
    from x265_amd.encoder import Encoder
    for key in ("1920x1080_f64_q32", "416x240_f8_q32"):
        size, f, q = key.split("_")
        w, h = (int(v) for v in size.split("x"))
        pics = clip(w, h, int(f[1:]))
        enc = Encoder(w, h, int(q[1:]), lanes=2, spin_us=200)
        try:
            with pytest.raises(XmiError):
                enc.collect()
            stream = b""
            for i, (y, u, v) in enumerate(pics):
                enc.submit(y, u, v, i == 0)
                if i == 1:
                    with pytest.raises(XmiError):
                        enc.submit(y, u, v, False)      # two pictures wait
                    with pytest.raises(XmiError):
                        enc.encode(y, u, v, False)
                if i:
                    stream += enc.collect()
            stream += enc.collect()
        finally:
            enc.close()
        assert len(stream) == GOLD[key]["bytes"] and hashlib.md5(stream).hexdigest() == GOLD[key]["md5"], key


def test_without_the_picture_hash_the_stream_is_the_one_of_the_reference_built_without_it():
    """xmi_encoder_picture_hash(e, 0): the reference's CHECK_SEI build option off (src/CMakeLists.txt:37); streams recorded from
    oracle/_ref/x265_refsse_nosei (tests/golden/encoder_md5_nosei.json), one- and two-step"""
    from x265_amd.encoder import Encoder
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5_nosei.json")))
    for key in sorted(gold):
        size, f, q = key.split("_")
        w, h = (int(v) for v in size.split("x"))
        pics = clip(w, h, int(f[1:]))
        for two_step in (False, True):
            enc = Encoder(w, h, int(q[1:]), picture_hash=False)
            try:
                stream = b""
                for i, (y, u, v) in enumerate(pics):
                    if two_step:
                        enc.submit(y, u, v, i == 0)
                        stream += enc.collect()
                    else:
                        stream += enc.encode(y, u, v, i == 0)
            finally:
                enc.close()
            assert len(stream) == gold[key]["bytes"] and hashlib.md5(stream).hexdigest() == gold[key]["md5"], (key, two_step)


def test_picture_hash_matches_the_reconstruction():
    """the SEI carries the MD5 of each reconstructed plane (src/encode.cpp:6012-6053): the writer's digest against hashlib"""
    from x265_amd.encoder import Encoder
    pics = clip(416, 240, 2)
    enc = Encoder(416, 240, 32)
    try:
        for i, (y, u, v) in enumerate(pics):
            au = enc.encode(y, u, v, i == 0)
            rec = enc.reconstruction()
            k = au.rfind(b"\x00\x00\x01\x50\x01\x84\x31\x00")
            assert k > 0
            # the payload is escaped like every NAL: strip emulation prevention bytes before comparing
            payload = au[k + 8:].replace(b"\x00\x00\x03", b"\x00\x00")
            want = b"".join(hashlib.md5(p.tobytes()).digest() for p in rec)
            assert payload[:48] == want
    finally:
        enc.close()


def test_errors():
    from x265_amd.capi import XmiError
    from x265_amd.encoder import Encoder
    pics = clip(416, 240, 1)
    enc = Encoder(416, 240, 32)
    try:
        with pytest.raises(XmiError):
            enc.encode(*pics[0], False)      # a P picture first
    finally:
        enc.close()
    with pytest.raises(XmiError):
        Encoder(416, 240, 77)
