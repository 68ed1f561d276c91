"""CPU: the host side of the encoder (x265_amd/csrc/host: decision loops, SAO decision, CABAC writer, headers) bound to the
CPU checker's primitives (oracle/libx265_analysis_check.so) writes the recorded streams byte for byte. The product binds the
same host source to the GPU (tests/test_encoder_gpu.py); this test holds the serial logic against the fixtures without one."""
import ctypes as C
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

LIB = os.path.join(ROOT, "oracle", "libx265_analysis_check.so")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_md5.json")))


def checker():
    if not os.path.exists(LIB):
        pytest.fail("oracle/libx265_analysis_check.so is not built (make -C oracle analysis_check)")
    lib = C.CDLL(LIB)
    lib.orc_encoder_create.restype = C.c_void_p
    lib.orc_encoder_create.argtypes = [C.c_int] * 4
    lib.orc_encoder_destroy.argtypes = [C.c_void_p]
    lib.orc_encoder_configure.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orc_encoder_encode.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_size_t]
    return lib


def encode(lib, w, h, frames, qp, lanes=1, speculate=1, picture_hash=True, refuse_pairs=0):
    from x265_amd.encoder import align_frame
    W, H = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    enc = lib.orc_encoder_create(W, H, qp, 0)
    lib.orc_encoder_configure(enc, lanes, 0)
    lib.orc_encoder_speculate.argtypes = [C.c_void_p, C.c_int]
    assert lib.orc_encoder_speculate(enc, speculate) == 0      # 2, the library's setting, costs the CPU harness several times the pixel work
    if not picture_hash:
        lib.orc_encoder_picture_hash.argtypes = [C.c_void_p, C.c_int]
        assert lib.orc_encoder_picture_hash(enc, 0) == 0
    if refuse_pairs:
        lib.orc_encoder_refuse_pairs.argtypes = [C.c_void_p, C.c_int]
        assert lib.orc_encoder_refuse_pairs(enc, refuse_pairs) == 0
    rng = np.random.default_rng(1234)
    out = np.zeros(W * H * 3 + 4096, np.uint8)
    stream, raw = b"", b""
    for t in range(frames):
        y, u, v = frame(w, h, t, rng)
        raw += y.tobytes() + u.tobytes() + v.tobytes()
        ya, ua, va = align_frame(y, u, v, W, H)
        n = lib.orc_encoder_encode(enc, ya.ctypes.data, ua.ctypes.data, va.ctypes.data, 2 if t == 0 else 1, out.ctypes.data, out.size)
        assert n > 0, n
        stream += bytes(out[:n])
    lib.orc_encoder_destroy(enc)
    return stream, raw


@pytest.mark.parametrize("every", [1, 3])
def test_an_evaluator_that_cannot_wait_for_the_block_to_the_left_makes_the_row_ask_again(every):
    """I pictures: the 8x8 block to the right of a block goes along with that block's request, evaluated after it; an evaluator
    may answer such a request with "no result" (best_mode -1). The stream is the recorded one whether every such request or
    every third is refused."""
    lib = checker()
    want = GOLD["416x240_f3_q22"]
    stream, _ = encode(lib, 416, 240, 3, 22, lanes=2, speculate=2, refuse_pairs=every)
    assert hashlib.md5(stream).hexdigest() == want["md5"]


@pytest.mark.parametrize("key,lanes,speculate", [("416x240_f8_q32", 1, 1), ("416x240_f3_q22", 2, 2), ("416x240_f8_q37", 1, 1)])
def test_host_encoder_writes_the_recorded_stream(key, lanes, speculate):
    lib = checker()
    size, f, q = key.split("_")
    w, h = (int(v) for v in size.split("x"))
    stream, raw = encode(lib, w, h, int(f[1:]), int(q[1:]), lanes, speculate)
    assert hashlib.md5(raw).hexdigest() == GOLD[key]["input_md5"]
    assert len(stream) == GOLD[key]["bytes"]
    assert hashlib.md5(stream).hexdigest() == GOLD[key]["md5"]


@needs_ref("libx265ref_pixel.so", how="make -C oracle ref")
@needs_ref("x265_refsse", how="make -C oracle enc")
@pytest.mark.parametrize("w,h,frames,qp,seed", [(128, 64, 4, 26, 4), (64, 192, 3, 30, 2), (64, 64, 3, 33, 9)])
def test_pictures_one_ctu_high_or_wide_against_the_reference_build(tmp_path, w, h, frames, qp, seed):
    """Degenerate geometries follow the reference's control flow to the letter: with one CTU row or column its deblocking
    gets to the last CTU only (hls_filters, src/Deblock.cpp:512), and a picture one CTU wide never stores the contexts the
    next row starts from (src/encode.cpp:5482, 5687), which stay zero."""
    import subprocess
    lib = checker()
    rng = np.random.default_rng(seed)
    pics = [frame(w, h, t, rng) for t in range(frames)]
    yuv, out = str(tmp_path / "in.yuv"), str(tmp_path / "ref.265")
    with open(yuv, "wb") as f:
        for p in pics:
            f.write(b"".join(c.tobytes() for c in p))
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "x265_refsse"), "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(frames), "-q", str(qp),
                    "-ip", "100", "-sceneChange", "0", "-t", "1"], check=True, capture_output=True, timeout=300)
    enc = lib.orc_encoder_create(w, h, qp, 0)
    buf = np.zeros(w * h * 3 + 4096, np.uint8)
    stream = b""
    for t, (y, u, v) in enumerate(pics):
        n = lib.orc_encoder_encode(enc, y.ctypes.data, u.ctypes.data, v.ctypes.data, 2 if t == 0 else 1, buf.ctypes.data, buf.size)
        assert n > 0, n
        stream += bytes(buf[:n])
    lib.orc_encoder_destroy(enc)
    assert stream == open(out, "rb").read()


def test_scan_tables_against_the_reference():
    """the horizontal / vertical / diagonal coefficient scans the writer derives equal the reference's tables
    (xg_ausScanIdx, xg_sigLastScan8x8, xg_sigLastScanCG32x32, src/table.cpp:268-511), read from its compiled table.cpp"""
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libx265ref_pixel.so"))
    lib = checker()
    lib.orc_syntax_scan.restype = C.POINTER(C.c_uint16)
    lib.orc_syntax_scan.argtypes = [C.c_int, C.c_int]
    lib.orc_syntax_group_scan.restype = C.POINTER(C.c_uint8)
    lib.orc_syntax_group_scan.argtypes = [C.c_int, C.c_int]
    table = ((C.POINTER(C.c_uint16) * 5) * 3).in_dll(ref, "xg_ausScanIdx")
    for kind in range(3):
        for log2 in (2, 3) if kind else (2, 3, 4, 5):      # the reference only ever walks the diagonal scan of 16x16 and 32x32 blocks
            n = 1 << (2 * log2)
            ours = [lib.orc_syntax_scan(kind, log2)[i] for i in range(n)]
            theirs = [table[kind][log2 - 1][i] for i in range(n)]
            assert ours == theirs, (kind, log2)
    cg8 = ((C.c_uint16 * 4) * 3).in_dll(ref, "xg_sigLastScan8x8")
    for kind in range(3):
        assert [lib.orc_syntax_group_scan(kind, 3)[i] for i in range(4)] == list(cg8[kind])
    cg32 = (C.c_uint16 * 64).in_dll(ref, "xg_sigLastScanCG32x32")
    assert [lib.orc_syntax_group_scan(0, 5)[i] for i in range(64)] == list(cg32)
    assert [lib.orc_syntax_group_scan(0, 4)[i] for i in range(16)] == [table[0][1][i] for i in range(16)]


def test_small_output_buffer_is_refused_and_nothing_is_written_beyond_it():
    """An access unit that does not fit the caller's buffer: the call fails, no byte beyond the capacity is touched, and the
    handle then wants an I picture (it holds no reference picture any more)."""
    from x265_amd.encoder import align_frame
    lib = checker()
    w, h, qp = 416, 240, 22
    W, H = 448, 256
    enc = lib.orc_encoder_create(W, H, qp, 0)
    lib.orc_encoder_configure(enc, 1, 0)
    rng = np.random.default_rng(1234)
    pics = [align_frame(*frame(w, h, t, rng), W, H) for t in range(2)]
    cap = 6000                                   # the I picture of this clip at QP 22 is about 27 KB
    out = np.full(cap + 4096, 0xA5, np.uint8)

    def call(pic, slice_type, capacity):
        return lib.orc_encoder_encode(enc, pic[0].ctypes.data, pic[1].ctypes.data, pic[2].ctypes.data, slice_type, out.ctypes.data, capacity)

    assert call(pics[0], 2, 100) == -100         # below the minimum: refused before anything runs
    n = call(pics[0], 2, cap)
    assert n == -102, n
    assert (out[cap:] == 0xA5).all(), "bytes beyond the capacity were written"
    assert call(pics[1], 1, out.size) == -101    # no reference picture after the failure
    big = np.zeros(W * H * 3 + 4096, np.uint8)
    n = lib.orc_encoder_encode(enc, pics[0][0].ctypes.data, pics[0][1].ctypes.data, pics[0][2].ctypes.data, 2, big.ctypes.data, big.size)
    assert n > cap
    lib.orc_encoder_destroy(enc)


def test_two_step_encode_gives_the_same_access_units():
    """submit / collect with one picture in flight: the access units equal encode()'s; the limits of the queue are errors and
    a collect into a buffer that is too small leaves the picture waiting."""
    from x265_amd.encoder import align_frame
    lib = checker()
    w, h, qp = 416, 240, 32
    W, H = 448, 256
    rng = np.random.default_rng(1234)
    pics = [align_frame(*frame(w, h, t, rng), W, H) for t in range(4)]
    lib.orc_encoder_submit.argtypes = [C.c_void_p] * 4 + [C.c_int]
    lib.orc_encoder_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    out = np.zeros(W * H * 3 + 8192, np.uint8)

    def args(t):
        return [p.ctypes.data for p in pics[t]] + [2 if t == 0 else 1]

    enc = lib.orc_encoder_create(W, H, qp, 0)
    lib.orc_encoder_configure(enc, 1, 0)
    plain = []
    for t in range(4):
        n = lib.orc_encoder_encode(enc, *args(t), out.ctypes.data, out.size)
        assert n > 0
        plain.append(bytes(out[:n]))
    lib.orc_encoder_destroy(enc)

    enc = lib.orc_encoder_create(W, H, qp, 0)
    lib.orc_encoder_configure(enc, 1, 0)
    assert lib.orc_encoder_collect(enc, out.ctypes.data, out.size) == -104       # nothing waits
    got = []
    assert lib.orc_encoder_submit(enc, *args(0)) == 0
    assert lib.orc_encoder_submit(enc, *args(1)) == 0
    assert lib.orc_encoder_submit(enc, *args(2)) == -103                         # two wait already
    assert lib.orc_encoder_encode(enc, *args(2), out.ctypes.data, out.size) == -103
    assert lib.orc_encoder_collect(enc, out.ctypes.data, 100) == -102            # too small: the picture keeps waiting
    for t in (2, 3):
        n = lib.orc_encoder_collect(enc, out.ctypes.data, out.size)
        assert n > 0
        got.append(bytes(out[:n]))
        assert lib.orc_encoder_submit(enc, *args(t)) == 0
    for _ in range(2):
        n = lib.orc_encoder_collect(enc, out.ctypes.data, out.size)
        assert n > 0
        got.append(bytes(out[:n]))
    lib.orc_encoder_destroy(enc)
    assert got == plain


def test_host_encoder_without_the_picture_hash_writes_the_stream_of_the_reference_built_without_it():
    """the reference's CHECK_SEI build option off (src/CMakeLists.txt:37): tests/golden/encoder_md5_nosei.json, written by
    oracle/_ref/x265_refsse_nosei (tools/enc_parity.py --nosei)"""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoder_md5_nosei.json")))
    lib = checker()
    for key in ("416x240_f8_q32", "416x240_f3_q22"):
        size, f, q = key.split("_")
        w, h = (int(v) for v in size.split("x"))
        stream, raw = encode(lib, w, h, int(f[1:]), int(q[1:]), picture_hash=False)
        assert hashlib.md5(raw).hexdigest() == gold[key]["input_md5"]
        assert len(stream) == gold[key]["bytes"] and hashlib.md5(stream).hexdigest() == gold[key]["md5"], key
