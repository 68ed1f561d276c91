"""GPU: the library's encoder behind the reference's own outer interface.

oracle/own_enc_hook.cpp defines the three functions of src/encode.cpp that the reference's front ends call - xEncInit,
xEncodeFrame (src/encode.cpp:5752), xEncFree - on xmi_encoder_*; oracle/Makefile (own_enc) links it with the reference's
command line (src/x265.cpp: option parsing, xFrameAligning, sceneChangeDetection, the -ip logic) into oracle/_ref/x265_own
and with its exported C API (src/interface.cpp:32-99, src/libx265.def:1-7) into oracle/_ref/libx265_own.so. None of the
reference's CTU loop, primitives, filters or entropy coder is in either link. Both must write the streams recorded in
tests/golden/encoder_md5.json."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_yuv import frame  # noqa: E402

from tests.prereq import REFDIR, needs_ref  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))
SEED = {416: 1234, 1920: 7, 3840: 1234}

CLI_CASES = ["416x240_f8_q32", "416x240_f3_q22", "416x240_f8_q32_cut6_scenechange", "416x240_f8_q32_cut5_scenechange", "1920x1080_f4_q32",
             "1920x1080_f3_q27"]


def parse(key):
    size, f, q = key.split("_")[:3]
    w, h = (int(v) for v in size.split("x"))
    cut = int(key.split("_cut")[1].split("_")[0]) if "_cut" in key else None
    return w, h, int(f[1:]), int(q[1:]), cut


def write_yuv(path, w, h, frames, cut=None):
    rng = np.random.default_rng(SEED[w])
    with open(path, "wb") as f:
        for t in range(frames):
            for p in frame(w, h, t, rng, cut):
                f.write(p.tobytes())
    return hashlib.md5(open(path, "rb").read()).hexdigest()


@needs_ref("x265_own", how="make -C oracle own_enc")
@pytest.mark.parametrize("key", CLI_CASES)
def test_the_reference_command_line_on_the_library_encoder(tmp_path, key):
    """416x240 and 1920x1080 are no multiples of 64: the command line's own xFrameAligning prepares the pictures; the two
    scene-cut clips run with its scene-change detection on (xSadN[3] of the drop-in table on the GPU)."""
    w, h, frames, qp, cut = parse(key)
    yuv, out = str(tmp_path / "in.yuv"), str(tmp_path / "own.265")
    assert write_yuv(yuv, w, h, frames, cut) == GOLD[key]["input_md5"], "the generator no longer makes the recorded input"
    res = subprocess.run([os.path.join(REFDIR, "x265_own"), "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(frames), "-q", str(qp),
                          "-ip", "100", "-sceneChange", "1" if cut is not None else "0", "-t", "1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    data = open(out, "rb").read()
    assert len(data) == GOLD[key]["bytes"], f"{len(data)} bytes, recorded {GOLD[key]['bytes']}"
    assert hashlib.md5(data).hexdigest() == GOLD[key]["md5"]


@needs_ref("x265_own_2160p", how="make -C oracle enc_2160p")
def test_the_reference_command_line_at_2160p(tmp_path):
    """the 3840x2160 clip of the benchmark (coded 3840x2176, beyond the reference's MAX_HEIGHT: its sources compiled against a
    copy of its headers with that one constant raised, as for x265_refsse_2160p)"""
    key = "3840x2160_f6_q32"
    w, h, frames, qp, _ = parse(key)
    yuv, out = str(tmp_path / "in.yuv"), str(tmp_path / "own.265")
    assert write_yuv(yuv, w, h, frames) == GOLD[key]["input_md5"]
    res = subprocess.run([os.path.join(REFDIR, "x265_own_2160p"), "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(frames), "-q", str(qp),
                          "-ip", "100", "-sceneChange", "0", "-t", "1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    data = open(out, "rb").read()
    assert (len(data), hashlib.md5(data).hexdigest()) == (GOLD[key]["bytes"], GOLD[key]["md5"])


class XFrame(C.Structure):      # inc/x265.h:148-155
    _fields_ = [("pucY", C.c_void_p), ("pucU", C.c_void_p), ("pucV", C.c_void_p), ("iStrideY", C.c_int), ("iStrideC", C.c_int), ("base", C.c_void_p)]


def api():
    lib = C.CDLL(os.path.join(REFDIR, "libx265_own.so"))
    lib.xIEncAlloc.restype = C.c_void_p
    lib.xIEncCheckParams.argtypes = [C.c_void_p]
    lib.xIEncInit.argtypes = [C.c_void_p]
    lib.xIEncFree.argtypes = [C.c_void_p]
    lib.xIEncSetParamInt.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.xIEncFrame.argtypes = [C.c_void_p, C.POINTER(XFrame), C.c_void_p, C.c_uint]
    return lib


@needs_ref("libx265_own.so", how="make -C oracle own_enc")
@pytest.mark.parametrize("key,stride_pad", [("416x240_f8_q32", 0), ("416x240_f8_q27", 32), ("1920x1080_f3_q37", 64)])
def test_the_exported_c_api_on_the_library_encoder(key, stride_pad):
    """xIEncAlloc -> xIEncSetParamInt("w" | "h" | "q" | "sis" | "st") -> xIEncCheckParams -> xIEncInit -> xIEncFrame per picture ->
    xIEncFree, as a user of the reference's DLL calls them; pictures at the coded size in an xFrame whose strides may exceed
    the width. In the first case two handles are alive at the same time (the second one encodes the same clip afterwards)."""
    from x265_amd.encoder import align_frame
    lib = api()
    w, h, frames, qp, _ = parse(key)
    W, H = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    rng = np.random.default_rng(SEED[w])
    pics = [align_frame(*frame(w, h, t, rng), W, H) for t in range(frames)]
    # (a handle is the reference's X265_t, a few hundred MB that xIEncFree does not give back: two only once)
    handles = [lib.xIEncAlloc() for _ in range(2 if stride_pad == 0 else 1)]
    streams = []
    try:
        for hd in handles:
            assert hd
            for name, val in ((b"w", W), (b"h", H), (b"q", qp), (b"sis", 0), (b"t", 1)):
                assert lib.xIEncSetParamInt(hd, name, val) == 0
            assert lib.xIEncSetParamInt(hd, b"no-such-option", 1) == -1
            lib.xIEncCheckParams(hd)
            lib.xIEncInit(hd)
        out = np.zeros(W * H * 3 + 4096, np.uint8)
        for hd in handles:
            stream = b""
            for t, (y, u, v) in enumerate(pics):
                sy, sc = W + stride_pad, W // 2 + stride_pad
                ys, us, vs = np.full((H, sy), 0xAA, np.uint8), np.full((H // 2, sc), 0xAA, np.uint8), np.full((H // 2, sc), 0xAA, np.uint8)
                ys[:, :W], us[:, :W // 2], vs[:, :W // 2] = y, u, v
                assert lib.xIEncSetParamInt(hd, b"st", 2 if t == 0 else 1) == 0      # xSliceType: SLICE_I = 2, SLICE_P = 1 (inc/x265.h:79-80)
                fr = XFrame(ys.ctypes.data, us.ctypes.data, vs.ctypes.data, sy, sc, None)
                n = lib.xIEncFrame(hd, C.byref(fr), out.ctypes.data, out.size)
                assert n > 0
                stream += bytes(out[:n])
            streams.append(stream)
    finally:
        for hd in handles:
            lib.xIEncFree(hd)
    for stream in streams:
        assert len(stream) == GOLD[key]["bytes"], f"{len(stream)} bytes, recorded {GOLD[key]['bytes']}"
        assert hashlib.md5(stream).hexdigest() == GOLD[key]["md5"]
