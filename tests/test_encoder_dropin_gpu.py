"""GPU: the reference encoder's own control flow (objects compiled from the reference sources into
oracle/_ref by `make -C oracle enc`) run with three sets of pixel primitives must write the same bitstream:
  x265_refsse  reference C++/SSE4 primitives (+ oracle for the NASM-only SATD/SSD entries)
  x265_oracle  all primitives from the CPU oracle
  x265_gpu     all primitives from libx265_amd through the drop-in table (per-call GPU kernels)
The CPU-sized configuration: 416x240, 8 frames, IPPP, CPU-sized. The md5s in tests/golden/encoder_md5.json were
written in the development container by the refsse build (tools/enc_parity.py --write)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from tests.prereq import needs_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
pytestmark = pytest.mark.gpu


def encode(variant, yuv, out, w, h, frames, qp, scene_change=0, threads=1):
    exe = os.path.join(REFDIR, f"x265_{variant}")
    cmd = [exe, "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(frames), "-q", str(qp), "-ip", "100",
           "-sceneChange", str(scene_change), "-t", str(threads)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), os.path.getsize(out)


@needs_ref("x265_gpu", how="make -C oracle enc")
@pytest.mark.parametrize("qp,frames", [(32, 8), (22, 3)])
def test_bitstream_identical_with_gpu_primitives(tmp_path, qp, frames):
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    assert hashlib.md5(open(yuv, "rb").read()).hexdigest() == want["input_md5"]
    got_o = encode("oracle", yuv, str(tmp_path / "o.265"), w, h, frames, qp)
    got_g = encode("gpu", yuv, str(tmp_path / "g.265"), w, h, frames, qp)
    assert got_o == (want["md5"], want["bytes"])
    assert got_g == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_me", how="make -C oracle trace")
@pytest.mark.parametrize("qp,frames", [(32, 8), (27, 8), (37, 8)])
def test_bitstream_identical_with_gpu_motion_search(tmp_path, qp, frames):
    """The reference's encoder with every xMotionSearch call served by the GPU library through the frame session
    (oracle/me_gpu_hook.cpp interposed; one xmi_frame_begin per picture, one request per call): the bitstream, which
    carries the MD5 of every reconstructed picture, must be the recorded one."""
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_me", yuv, str(tmp_path / "m.265"), w, h, frames, qp)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_planes", how="make -C oracle trace")
@pytest.mark.parametrize("qp,frames", [(32, 8), (22, 3)])
def test_bitstream_identical_with_gpu_interpolation_planes(tmp_path, qp, frames):
    """The reference's encoder with xPredInterLumaBlk replaced by copies out of the GPU-built quarter-pel planes
    (oracle/planes_gpu_hook.cpp interposed; motion search, merge candidates and motion compensation all read them)."""
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_planes", yuv, str(tmp_path / "p.265"), w, h, frames, qp)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_deblock", how="make -C oracle trace")
@pytest.mark.parametrize("qp,frames", [(32, 8), (37, 8), (22, 3)])
def test_bitstream_identical_with_gpu_deblocking(tmp_path, qp, frames):
    """The reference's encoder with xDeblockFrame replaced by xmi_deblock_frame (oracle/deblock_gpu_hook.cpp; the original
    is never called): the filtered reconstruction is the next picture's reference and is hashed into the stream's MD5
    SEI, so the recorded bitstream means identical planes for every picture."""
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_deblock", yuv, str(tmp_path / "d.265"), w, h, frames, qp)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_bs", how="make -C oracle trace")
@pytest.mark.parametrize("qp,frames", [(32, 8), (37, 8), (22, 3)])
def test_bitstream_identical_with_gpu_boundary_strengths(tmp_path, qp, frames):
    """The reference's encoder with ff_hevc_deblocking_boundary_strengths replaced by a recorder of picture-level maps and
    one xmi_deblock_strengths per picture (oracle/bs_gpu_hook.cpp; the original is never called, the arrays are poisoned
    before the library fills them): the reference's own filter then reads GPU-derived strengths, and the stream with its
    reconstruction hashes is the recorded one."""
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_bs", yuv, str(tmp_path / "b.265"), w, h, frames, qp)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_sao", how="make -C oracle trace")
@pytest.mark.parametrize("qp,frames", [(32, 8), (37, 8), (22, 3)])
def test_bitstream_identical_with_gpu_sao_statistics(tmp_path, qp, frames):
    """The reference's encoder with calcSaoStatsCu served from one GPU launch per picture (oracle/sao_gpu_hook.cpp; the
    original is never called): the SAO parameters decided from the tables are coded into the stream and shape the
    reconstruction, so the recorded bitstream means identical tables."""
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_sao", yuv, str(tmp_path / "s.265"), w, h, frames, qp)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_intra", how="make -C oracle trace")
@pytest.mark.parametrize("qp,frames", [(32, 8), (22, 3)])
def test_bitstream_identical_with_gpu_intra_search(tmp_path, qp, frames):
    """The reference's encoder with every intra luma mode search answered by the GPU (oracle/intra_gpu_hook.cpp: the 35
    SATD calls of a search return the numbers of one xmi_intra_search on the samples the encoder had before padding;
    the reference's SATD code does not run for them): the recorded bitstream, i.e. the same 35 SATDs everywhere."""
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_intra", yuv, str(tmp_path / "i.265"), w, h, frames, qp)
    assert got == (want["md5"], want["bytes"])
    # the hook reports how many searches the GPU answered: all of them, not a silent pass-through
    exe = os.path.join(REFDIR, "x265_gpu_intra")
    res = subprocess.run([exe, "-i", yuv, "-o", str(tmp_path / "j.265"), "-w", str(w), "-h", str(h), "-f", "1", "-q", str(qp), "-ip", "100",
                          "-sceneChange", "0", "-t", "1"], capture_output=True, text=True, timeout=600)
    import re
    m = re.search(r"intra_gpu_hook: (\d+) searches on the GPU, (\d+) other SATD calls", res.stderr)
    assert m and int(m.group(1)) > 2000, res.stderr[-500:]


@needs_ref("x265_gpu_all", how="make -C oracle trace")
def test_bitstream_identical_with_every_gpu_service(tmp_path):
    """Motion search from the frame session AND every pixel primitive (SAD / SATD / SSD tables, xSubDct, xQuant,
    xDeQuant, xIDctAdd, xPredInterLumaBlk, xPredInterChromaBlk, xEncIntraPred) from libx265_amd_dropin.so, which sits ahead of the reference's library in the link order and
    so replaces its definitions: the reference's control flow with no CPU pixel arithmetic left in the luma/transform path."""
    w, h, frames, qp = 416, 240, 8, 32
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_all", yuv, str(tmp_path / "a.265"), w, h, frames, qp)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_me", how="make -C oracle trace")
def test_gop_segments_on_two_ranks_with_gpu_motion_search():
    """tools/encode_segments.py, two ranks (gloo rendezvous, both on this box's GPU), each encoding its GOP with the GPU
    motion search: every gathered segment is the stream recorded for it (tests/golden/segment_md5.json)."""
    from tests.test_segments_gloo import run_segment_encode
    res = run_segment_encode("gpu_me", 2)
    assert res["parity"] == "identical" and res["ranks"] == 2


@needs_ref("x265_gpu_me_2160p", how="make -C oracle enc_2160p")
def test_2160p_segments_on_two_ranks_with_gpu_motion_search():
    """The 3840x2160 segment fixture (tests/golden/segment_md5_2160p.json, recorded by the CPU build of the same
    height-patched sources): two ranks, one 2-frame segment each, every motion search on the GPU."""
    from tests.test_segments_gloo import UHD, run_segment_encode
    res = run_segment_encode("gpu_me_2160p", 2, clip=UHD)
    assert res["parity"] == "identical" and res["ranks"] == 2 and set(res["segments"]) == {"0-2", "2-4"}


@needs_ref("x265_gpu_all", how="make -C oracle trace")
@pytest.mark.parametrize("variant", ["gpu", "gpu_all"])
def test_scene_change_detection_on_gpu_sad(tmp_path, variant):
    """The reference's scene-change detection (sceneChangeDetection, src/preProcess.cpp:104: 32x32 SADs of the source
    against the previous reconstruction through xSadN[3]) with the GPU's SAD table, on a clip with a scene cut at
    picture 6: the cut is detected, an I picture is inserted, and the stream is the one the CPU build writes."""
    w, h, frames, qp = 416, 240, 8, 32
    yuv = str(tmp_path / "cut.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv, "6"])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))["416x240_f8_q32_cut6_scenechange"]
    assert hashlib.md5(open(yuv, "rb").read()).hexdigest() == want["input_md5"]
    got = encode(variant, yuv, str(tmp_path / "c.265"), w, h, frames, qp, scene_change=1)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_mt", how="make -C oracle enc")
def test_bitstream_identical_with_row_threads_on_gpu_primitives(tmp_path):
    """The reference's own OpenMP row threads (src/encode.cpp:5865; the build of x265_refsse_mt) calling the GPU per-call
    primitives concurrently, four rows at a time: the per-call entry points serialise on their staging buffers, and the
    stream is the single-thread one."""
    w, h, frames, qp = 416, 240, 8, 32
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    got = encode("gpu_mt", yuv, str(tmp_path / "t.265"), w, h, frames, qp, threads=4)
    assert got == (want["md5"], want["bytes"])


@needs_ref("x265_gpu_me", how="make -C oracle trace")
@pytest.mark.parametrize("key,variants", [("1920x1080_f4_q32", ("gpu_me", "gpu_planes", "gpu_deblock", "gpu_sao", "gpu_bs")), ("1920x1080_f3_q22", ("gpu_me", "gpu_deblock")),
                                          ("1920x1080_f3_q27", ("gpu_me", "gpu_deblock")), ("1920x1080_f3_q37", ("gpu_me", "gpu_deblock"))])
def test_bitstream_identical_at_1080p_with_gpu_services(tmp_path, key, variants):
    """The geometry the headline metric is quoted on (1920x1080, coded 1920x1088, 510 CTUs; generator seed 7): the
    reference's encoder with the GPU motion search / quarter-pel planes / deblocking / SAO statistics writes the stream
    recorded from the CPU build."""
    w, h = 1920, 1080
    frames, qp = int(key.split("_")[1][1:]), int(key.split("_")[2][1:])
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "7", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[key]
    assert hashlib.md5(open(yuv, "rb").read()).hexdigest() == want["input_md5"]
    for v in variants:
        assert encode(v, yuv, str(tmp_path / f"{v}.265"), w, h, frames, qp) == (want["md5"], want["bytes"]), f"x265_{v}"


@needs_ref("x265_gpu_chroma", how="make -C oracle trace")
@pytest.mark.parametrize("qp,frames", [(32, 8), (22, 3)])
def test_bitstream_identical_with_gpu_chroma_search(tmp_path, qp, frames):
    """The reference's encoder with every intra chroma mode search answered by the GPU (oracle/chroma_gpu_hook.cpp: the ten
    SATD calls of a search return the numbers of one xmi_intra_chroma_search; the reference's SATD code does not run for
    them): the recorded bitstream, i.e. the same SATDs in every search."""
    w, h = 416, 240
    yuv = str(tmp_path / "in.yuv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), "1234", yuv])
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))[f"416x240_f{frames}_q{qp}"]
    exe = os.path.join(REFDIR, "x265_gpu_chroma")
    out = str(tmp_path / "c.265")
    res = subprocess.run([exe, "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(frames), "-q", str(qp), "-ip", "100",
                          "-sceneChange", "0", "-t", "1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-1000:]
    assert (hashlib.md5(open(out, "rb").read()).hexdigest(), os.path.getsize(out)) == (want["md5"], want["bytes"])
    import re
    m = re.search(r"chroma_gpu_hook: (\d+) searches on the GPU", res.stderr)
    assert m and int(m.group(1)) >= 200, res.stderr[-500:]
