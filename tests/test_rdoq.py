"""Rate-distortion optimised quantisation ("RdoQ"). PARITY UNPINNED BY CONSTRUCTION: the reference compiles its RdoQ.cpp out
and defines no level decision (src/RdoQ.cpp:32, inc/RdoQ.h:55), so nothing here compares with the reference. What is
checked: the CPU checker's statement of the algorithm behaves like a quantiser should (CPU tests), the library's table
builders equal the checker's restatement of src/RdoQ.cpp:67-305 (CPU), and the HIP kernel reproduces the checker bit for
bit through the C ABI (GPU tests)."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O

Q6 = [26214, 23302, 20560, 18396, 16384, 14564]


class Tables(C.Structure):      # xmi_rdoq_tables / orc_rdoq_tables
    _fields_ = [("sig_cg", C.c_int32 * 2 * 2), ("sig", C.c_int32 * 2 * 27), ("last_x", C.c_int32 * 12), ("last_y", C.c_int32 * 12),
                ("one", C.c_int32 * 2 * 16), ("abs", C.c_int32 * 2 * 4), ("cbf", C.c_int32 * 2)]


class Lambda(C.Structure):
    _fields_ = [("mantissa", C.c_uint32), ("shift", C.c_int32)]


def orc():
    lib = O.load()
    lib.orc_rdoq_est_bits.restype = None
    lib.orc_rdoq_est_bits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Tables)]
    lib.orc_rdoq_lambda.restype = None
    lib.orc_rdoq_lambda.argtypes = [C.c_double, C.c_uint, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    lib.orc_rdoq_quant.restype = C.c_uint32
    lib.orc_rdoq_quant.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.POINTER(Tables), C.c_uint32, C.c_int32]
    return lib


def product():
    from x265_amd import capi
    L = capi.load()
    L.xmi_rdoq_est_bits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Tables)]
    L.xmi_rdoq_initial_contexts.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.xmi_rdoq_lambda_of.argtypes = [C.c_double, C.c_int, C.c_int, C.POINTER(Lambda)]
    L.xmi_rdoq_quant_batch_dev.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int] + [C.c_void_p] * 3
    L.xmi_tu_chain_rdoq_batch_dev.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int] + [C.c_void_p] * 5
    return L


def lam_of(slice_type, qp):
    return (0.57 if slice_type == O.SLICE_I else 0.578) * 2.0 ** ((qp - 12) / 3.0)      # src/encode.cpp:5348, 5406


def coefficients(rng, n, count, kind):
    """transform-like blocks: energy falling off with frequency; `kind` scales it from mostly-zero to dense"""
    fy, fx = np.mgrid[0:n, 0:n]
    env = {"sparse": 40.0, "medium": 400.0, "dense": 4000.0}[kind] / (1.0 + fx + fy) ** 1.5
    c = rng.laplace(0.0, 1.0, (count, n, n)) * env
    c[:, 0, 0] *= 4
    return np.clip(np.rint(c), -32768, 32767).astype(np.int16)


def contexts(L, slice_type, qp):
    ctx = np.zeros(256, np.uint8)
    assert L.xmi_rdoq_initial_contexts(slice_type, qp, O.p(ctx)) == 0
    return ctx


def plain_levels(coef, n, qp):
    lg = int(np.log2(n))
    qbits = 14 + qp // 6 + (15 - 8 - lg)
    return np.minimum((np.abs(coef.astype(np.int64)) * Q6[qp % 6] + (1 << (qbits - 1))) >> qbits, 32767)


# ---------------------------------------------------------------------------------------------------------- CPU
def test_table_builders_and_lambda_equal_the_checker():
    L, lib = product(), orc()
    rng = np.random.default_rng(5)
    for slice_type, qp in ((O.SLICE_P, 32), (O.SLICE_I, 22), (O.SLICE_P, 51), (O.SLICE_I, 0)):
        ctx = contexts(L, slice_type, qp)
        noisy = ctx.copy()
        noisy[:184] = rng.integers(0, 126, 184)
        for c in (ctx, noisy):
            for lg in (2, 3, 4, 5):
                for luma in (1, 0):
                    a, b = Tables(), Tables()
                    assert L.xmi_rdoq_est_bits(O.p(c), lg, luma, C.byref(a)) == 0
                    lib.orc_rdoq_est_bits(O.p(c), lg, luma, C.byref(b))
                    assert bytes(a) == bytes(b), (slice_type, qp, lg, luma)
                    assert a.sig[0][0] > 0 and a.last_x[0] > 0
                lam = Lambda()
                m, s = C.c_uint32(), C.c_int32()
                assert L.xmi_rdoq_lambda_of(lam_of(slice_type, qp), qp, lg, C.byref(lam)) == 0
                lib.orc_rdoq_lambda(lam_of(slice_type, qp), qp, lg, C.byref(m), C.byref(s))
                assert (lam.mantissa, lam.shift) == (m.value, s.value) and 0 < lam.mantissa < 2 ** 30


def test_without_rate_the_decision_is_rounding_up_to_ties_and_with_a_high_price_nothing_is_coded():
    L, lib = product(), orc()
    rng = np.random.default_rng(6)
    for n in (4, 8, 16, 32):
        t = Tables()
        L.xmi_rdoq_est_bits(O.p(contexts(L, O.SLICE_P, 30)), int(np.log2(n)), 1, C.byref(t))
        for blk in coefficients(rng, n, 6, "medium"):
            lev = np.zeros((n, n), np.int16)
            s = lib.orc_rdoq_quant(O.p(lev), O.p(blk), n, n, 30, 1, 0, C.byref(t), 0, 0)
            # at a price of zero only the error counts: the rounded level, or - where the error in sixteenths of a level is the
            # same for both - the smaller neighbour (the cheaper candidate is kept on a tie)
            mx, a = plain_levels(blk, n, 30), np.abs(lev.astype(np.int64))
            assert ((a == mx) | (a == mx - 1)).all() and (a == mx).mean() > 0.9 and (np.sign(lev) * np.sign(blk) >= 0).all()
            assert s == a.sum()
            s = lib.orc_rdoq_quant(O.p(lev), O.p(blk), n, n, 30, 1, 0, C.byref(t), (1 << 30) - 1, 0)
            assert s == 0 and not lev.any()


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_levels_are_the_rounded_level_one_less_or_zero_and_cost_less_to_code(n):
    """every level is the rounded one, one less, or zero, with the coefficient's sign; over a batch the decision codes fewer
    levels than rounding and no more than the reference's dead-zone quantiser leaves at a price of zero"""
    L, lib = product(), orc()
    rng = np.random.default_rng(7 + n)
    qp, slice_type = 32, O.SLICE_P
    t = Tables()
    L.xmi_rdoq_est_bits(O.p(contexts(L, slice_type, qp)), int(np.log2(n)), 1, C.byref(t))
    lam = Lambda()
    L.xmi_rdoq_lambda_of(lam_of(slice_type, qp), qp, int(np.log2(n)), C.byref(lam))
    total, total_round = 0, 0
    for kind in ("sparse", "medium", "dense"):
        for blk in coefficients(rng, n, 10, kind):
            lev = np.zeros((n, n), np.int16)
            s = lib.orc_rdoq_quant(O.p(lev), O.p(blk), n, n, qp, 1, 0, C.byref(t), lam.mantissa, lam.shift)
            mx = plain_levels(blk, n, qp)
            a = np.abs(lev.astype(np.int64))
            assert ((a == mx) | (a == mx - 1) | (a == 0)).all()
            assert (np.sign(lev) * np.sign(blk) >= 0).all() and s == a.sum()
            total += int(s)
            total_round += int(mx.sum())
    assert 0 < total < total_round


def test_strided_blocks_and_scans():
    """levels are written with the coefficients' stride (as xQuant does); the three scans give different decisions on the
    same block only through the context choice, and all of them stay within the candidate set"""
    L, lib = product(), orc()
    rng = np.random.default_rng(9)
    n, qp = 8, 27
    t = Tables()
    L.xmi_rdoq_est_bits(O.p(contexts(L, O.SLICE_I, qp)), 3, 1, C.byref(t))
    lam = Lambda()
    L.xmi_rdoq_lambda_of(lam_of(O.SLICE_I, qp), qp, 3, C.byref(lam))
    blk = coefficients(rng, n, 1, "medium")[0]
    wide = np.zeros((n, 64), np.int16)
    wide[:, :n] = blk
    for scan in (0, 1, 2):
        a, b = np.zeros((n, n), np.int16), np.full((n, 64), 77, np.int16)
        sa = lib.orc_rdoq_quant(O.p(a), O.p(blk), n, n, qp, 1, scan, C.byref(t), lam.mantissa, lam.shift)
        sb = lib.orc_rdoq_quant(O.p(b), O.p(wide), 64, n, qp, 1, scan, C.byref(t), lam.mantissa, lam.shift)
        assert sa == sb and np.array_equal(a, b[:, :n]) and (b[:, n:] == 77).all()


# ---------------------------------------------------------------------------------------------------------- GPU
def run_gpu(L, torch, n, coef, qps, t, lams, scan, luma):
    cnt = coef.shape[0]
    d_coef = torch.from_numpy(coef).cuda()
    d_qp = torch.from_numpy(qps).cuda()
    d_t = torch.from_numpy(np.frombuffer(bytes(t), np.uint8).copy()).cuda()
    d_lam = torch.from_numpy(np.frombuffer(bytes(lams), np.uint8).copy()).cuda()
    d_lev = torch.full((cnt, n, n), 99, dtype=torch.int16, device="cuda")
    d_sum = torch.full((cnt,), 12345, dtype=torch.int32, device="cuda")
    rc = L.xmi_rdoq_quant_batch_dev(n, cnt, d_coef.data_ptr(), d_qp.data_ptr(), None, d_t.data_ptr(), d_lam.data_ptr(), scan, luma, d_lev.data_ptr(),
                                    d_sum.data_ptr(), None)
    assert rc == 0, L.xmi_last_error()
    torch.cuda.synchronize()
    return d_lev.cpu().numpy(), d_sum.cpu().numpy().astype(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 8, 16, 32])
@pytest.mark.parametrize("luma,slice_type", [(1, O.SLICE_P), (0, O.SLICE_I)])
def test_kernel_equals_the_checker(n, luma, slice_type):
    import torch
    from x265_amd import capi
    L, lib = product(), orc()
    capi.check(L.xmi_init(0))
    rng = np.random.default_rng(100 + n + luma)
    lg = int(np.log2(n))
    lams = (Lambda * 52)()
    for qp in range(52):
        L.xmi_rdoq_lambda_of(lam_of(slice_type, qp), qp, lg, C.byref(lams[qp]))
    ctx = contexts(L, slice_type, 30)
    ctx[:184] = np.where(rng.random(184) < 0.5, ctx[:184], rng.integers(0, 126, 184)).astype(np.uint8)      # states a running encode would have
    t = Tables()
    L.xmi_rdoq_est_bits(O.p(ctx), lg, luma, C.byref(t))
    per = 40 if n < 32 else 16
    coef = np.concatenate([coefficients(rng, n, per, k) for k in ("sparse", "medium", "dense")] +
                          [np.zeros((2, n, n), np.int16), np.full((1, n, n), 32767, np.int16), np.full((1, n, n), -32768, np.int16)])
    coef[-4, n - 1, n - 1] = 3000      # a lone coefficient at the far corner of an otherwise empty block
    qps = rng.integers(0, 52, coef.shape[0]).astype(np.uint8)
    qps[-2:] = (0, 51)
    for scan in ((0, 1, 2) if n <= 8 else (0,)):
        got_lev, got_sum = run_gpu(L, torch, n, coef, qps, t, lams, scan, luma)
        nonzero_blocks = 0
        for i in range(coef.shape[0]):
            want = np.zeros((n, n), np.int16)
            s = lib.orc_rdoq_quant(O.p(want), O.p(coef[i]), n, n, int(qps[i]), luma, scan, C.byref(t), lams[qps[i]].mantissa, lams[qps[i]].shift)
            assert got_sum[i] == s, (i, scan, int(qps[i]))
            np.testing.assert_array_equal(got_lev[i], want, err_msg=f"block {i} scan {scan} qp {qps[i]}")
            nonzero_blocks += s > 0
        assert nonzero_blocks > coef.shape[0] // 3


@pytest.mark.gpu
def test_chain_with_the_decision_equals_the_checker_chain():
    """xSubDct -> decision -> xDeQuant -> xIDctAdd on the device against the same four steps of the checker"""
    import torch
    from x265_amd import capi
    L, lib = product(), orc()
    capi.check(L.xmi_init(0))
    rng = np.random.default_rng(31)
    for n, qp in ((8, 27), (16, 32), (32, 37), (4, 22)):
        lg = int(np.log2(n))
        cnt = 24
        pix = rng.integers(0, 256, (cnt, n, n), dtype=np.uint8)
        pred = np.clip(pix.astype(int) + rng.integers(-20, 21, pix.shape) + (rng.random(pix.shape) < 0.05) * 60, 0, 255).astype(np.uint8)
        lams = (Lambda * 52)()
        for q in range(52):
            L.xmi_rdoq_lambda_of(lam_of(O.SLICE_P, q), q, lg, C.byref(lams[q]))
        t = Tables()
        L.xmi_rdoq_est_bits(O.p(contexts(L, O.SLICE_P, qp)), lg, 1, C.byref(t))
        d = {k: torch.from_numpy(v).cuda() for k, v in (("pix", pix), ("pred", pred), ("qp", np.full(cnt, qp, np.uint8)), ("flags", np.zeros(cnt, np.uint8)))}
        d_t = torch.from_numpy(np.frombuffer(bytes(t), np.uint8).copy()).cuda()
        d_lam = torch.from_numpy(np.frombuffer(bytes(lams), np.uint8).copy()).cuda()
        d_work = torch.zeros((cnt, n, n), dtype=torch.int16, device="cuda")
        d_lev = torch.zeros((cnt, n, n), dtype=torch.int16, device="cuda")
        d_rec = torch.zeros((cnt, n, n), dtype=torch.uint8, device="cuda")
        d_sum = torch.zeros(cnt, dtype=torch.int32, device="cuda")
        rc = L.xmi_tu_chain_rdoq_batch_dev(n, cnt, d["pix"].data_ptr(), d["pred"].data_ptr(), d["qp"].data_ptr(), d["flags"].data_ptr(), d_t.data_ptr(),
                                           d_lam.data_ptr(), 0, 1, d_work.data_ptr(), d_lev.data_ptr(), d_rec.data_ptr(), d_sum.data_ptr(), None)
        assert rc == 0, L.xmi_last_error()
        torch.cuda.synchronize()
        lev, rec, sums = d_lev.cpu().numpy(), d_rec.cpu().numpy(), d_sum.cpu().numpy()
        coded = 0
        for i in range(cnt):
            coef = np.zeros((n, n), np.int16)
            lib.orc_sub_dct(O.p(coef), O.p(pix[i]), n, O.p(pred[i]), n, n, O.MODE_INVALID)
            want = np.zeros((n, n), np.int16)
            s = lib.orc_rdoq_quant(O.p(want), O.p(coef), n, n, qp, 1, 0, C.byref(t), lams[qp].mantissa, lams[qp].shift)
            work = np.zeros((n, n), np.int16)
            lib.orc_dequant(O.p(work), O.p(want), n, qp, n, n)
            wrec = np.zeros((n, n), np.uint8)
            lib.orc_idct_add(O.p(wrec), n, O.p(work), n, O.p(pred[i]), n, O.MODE_INVALID)
            assert sums[i] == s
            np.testing.assert_array_equal(lev[i], want)
            np.testing.assert_array_equal(rec[i], wrec)
            coded += s > 0
        assert coded > 0


@pytest.mark.gpu
def test_errors():
    from x265_amd import capi
    L = product()
    capi.check(L.xmi_init(0))
    assert L.xmi_rdoq_quant_batch_dev(12, 1, 8, 8, None, 8, 8, 0, 1, 8, 8, None) == -1
    assert L.xmi_rdoq_quant_batch_dev(8, 1, 8, 8, None, 8, 8, 3, 1, 8, 8, None) == -1
    assert L.xmi_rdoq_quant_batch_dev(8, 0, 8, 8, None, 8, 8, 0, 1, 8, 8, None) == 0      # an empty batch
    assert L.xmi_rdoq_est_bits(None, 3, 1, None) == -1


# ------------------------------------------------------------------------------------------------- in the encoder
def harness_stream(w, h, frames, qp, rdoq):
    """the product's host code on the CPU checker's primitives (oracle/libx265_analysis_check.so), level decision on or off"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from gen_yuv import frame
    from x265_amd.encoder import align_frame
    lib = C.CDLL(os.path.join(root, "oracle", "libx265_analysis_check.so"))
    lib.orc_encoder_create.restype = C.c_void_p
    lib.orc_encoder_create.argtypes = [C.c_int] * 4
    lib.orc_encoder_destroy.argtypes = [C.c_void_p]
    lib.orc_encoder_rdoq.argtypes = [C.c_void_p, C.c_int]
    lib.orc_encoder_encode.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_size_t]
    W, H = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    enc = lib.orc_encoder_create(W, H, qp, 0)
    assert lib.orc_encoder_rdoq(enc, int(rdoq)) == 0
    rng = np.random.default_rng(1234)
    out = np.zeros(W * H * 3 + 4096, np.uint8)
    aus = []
    for t in range(frames):
        y, u, v = align_frame(*frame(w, h, t, rng), W, H)
        n = lib.orc_encoder_encode(enc, y.ctypes.data, u.ctypes.data, v.ctypes.data, 2 if t == 0 else 1, out.ctypes.data, out.size)
        assert n > 0, n
        aus.append(bytes(out[:n]))
    lib.orc_encoder_destroy(enc)
    return aus


def test_encoder_with_the_decision_on_the_cpu_harness():
    """The P pictures change (the decision weighs squared error against bits at the decision loop's own Lagrange multiplier, where
    the reference's quantiser rounds with a fixed dead zone: on this noisy clip it spends fewer bits at QP 37 and more at QP 22),
    the I picture is untouched, and the stream stays what the plain encoder writes when the switch is off"""
    plain, rd = harness_stream(416, 240, 4, 37, False), harness_stream(416, 240, 4, 37, True)
    assert plain[0] == rd[0]
    assert all(a != b for a, b in zip(plain[1:], rd[1:]))
    assert sum(map(len, rd[1:])) < sum(map(len, plain[1:]))
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoder_md5.json")))["416x240_f8_q27"]
    assert hashlib.md5(b"".join(harness_stream(416, 240, 8, 27, False))).hexdigest() == gold["md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,frames,qp", [(416, 240, 4, 27), (416, 240, 3, 22), (1920, 1080, 3, 32)])
def test_library_encoder_with_the_decision_equals_the_cpu_harness(w, h, frames, qp):
    """xmi_encoder_rdoq: the library's stream is the one its host code writes on the CPU checker's primitives with the checker's
    statement of the decision, access unit by access unit; the picture hash in every access unit is the hash of the
    reconstruction the encoder holds (a decoder of the stream would have to arrive at the same picture)."""
    import hashlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_yuv import frame
    from x265_amd.encoder import Encoder
    want = harness_stream(w, h, frames, qp, True)
    rng = np.random.default_rng(1234)
    enc = Encoder(w, h, qp, rdoq=True)
    try:
        for t in range(frames):
            au = enc.encode(*frame(w, h, t, rng), t == 0)
            assert len(au) == len(want[t]) and au == want[t], f"picture {t}"
            rec = enc.reconstruction()
            k = au.rfind(b"\x00\x00\x01\x50\x01\x84\x31\x00")
            payload = au[k + 8:].replace(b"\x00\x00\x03", b"\x00\x00")
            assert payload[:48] == b"".join(hashlib.md5(p.tobytes()).digest() for p in rec)
    finally:
        enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_1080p_qp_sweep_with_the_decision(qp):
    """1920x1080 at every QP of the published sweep with the decision on: the P pictures differ from the plain quantiser's, every
    access unit's picture hash is the hash of the encoder's reconstruction, and switching the decision off again on a new
    handle gives back the recorded stream of the reference (tests/golden/encoder_md5.json)."""
    import hashlib
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from gen_yuv import frame
    from x265_amd.encoder import Encoder
    gold = json.load(open(os.path.join(root, "tests", "golden", "encoder_md5.json")))
    key = next(k for k in sorted(gold) if k.startswith("1920x1080_f") and k.endswith(f"_q{qp}") and "f64" not in k)
    frames = int(key.split("_")[1][1:])
    streams = {}
    for on in (True, False):
        rng = np.random.default_rng(7)
        enc = Encoder(1920, 1080, qp, rdoq=on)
        out = []
        try:
            for t in range(frames):
                au = enc.encode(*frame(1920, 1080, t, rng), t == 0)
                rec = enc.reconstruction()
                k = au.rfind(b"\x00\x00\x01\x50\x01\x84\x31\x00")
                payload = au[k + 8:].replace(b"\x00\x00\x03", b"\x00\x00")
                assert payload[:48] == b"".join(hashlib.md5(p.tobytes()).digest() for p in rec), (on, t)
                out.append(au)
        finally:
            enc.close()
        streams[on] = out
    plain = b"".join(streams[False])
    assert len(plain) == gold[key]["bytes"] and hashlib.md5(plain).hexdigest() == gold[key]["md5"]
    assert streams[True][0] == streams[False][0]                  # the I picture keeps the reference's quantiser
    assert streams[True][1:] != streams[False][1:]                # the P pictures take the decision's levels
