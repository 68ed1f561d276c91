"""GPU: batched block metrics, the TU chain and the per-call drop-ins against the oracle and the golden
vectors produced by the reference's compiled code. Bit-exact, through the C ABI."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from x265_amd import capi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def L():
    lib = capi.load()
    capi.check(lib.xmi_init(0))
    return lib


@pytest.fixture(scope="module")
def orc():
    return O.load()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def metric_gpu(L, metric, n, A, B):
    da, db = dev(A), dev(B)
    out = torch.zeros(len(A), dtype=torch.int32, device="cuda")
    capi.check(L.xmi_block_metric_batch_dev(metric, n, len(A), da.data_ptr(), db.data_ptr(), out.data_ptr(), None))
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.uint32)


def test_block_metrics_golden(L, golden_dir):
    g = np.load(os.path.join(golden_dir, "sad_satd.npz"))
    for n in (4, 8, 16, 32, 64):
        A, B = g[f"a{n}"], g[f"b{n}"]
        np.testing.assert_array_equal(metric_gpu(L, capi.METRIC_SAD, n, A, B), g[f"sad{n}"])
        np.testing.assert_array_equal(metric_gpu(L, capi.METRIC_SATD_HM, n, A, B), g[f"satd_hm{n}"])
        d = A.astype(np.int64) - B.astype(np.int64)
        np.testing.assert_array_equal(metric_gpu(L, capi.METRIC_SSD, n, A, B), (d * d).sum(axis=(1, 2)))
    np.testing.assert_array_equal(metric_gpu(L, capi.METRIC_SATD_TAB, 32, g["a32"], g["b32"]), g["satd_tab3_32"])
    # NASM-only 4x4 SATD == the reference's compiled xCalcHADs4x4 (the one point where that family is pinned)
    np.testing.assert_array_equal(metric_gpu(L, capi.METRIC_SATD_X264, 4, g["a4"], g["b4"]), g["satd_hm4"])


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_block_metrics_random_vs_oracle(L, orc, n):
    rng = np.random.default_rng(n)
    cnt = 777 if n <= 16 else 130     # ragged: not a multiple of the lanes per workgroup
    A = rng.integers(0, 256, (cnt, n, n), dtype=np.uint8)
    B = np.clip(A.astype(int) + rng.integers(-40, 41, A.shape), 0, 255).astype(np.uint8)
    B[::7] = rng.integers(0, 256, B[::7].shape, dtype=np.uint8)
    fns = [(capi.METRIC_SAD, orc.orc_sad), (capi.METRIC_SSD, orc.orc_ssd), (capi.METRIC_SATD_HM, orc.orc_satd_hm)]
    if n <= 16:
        fns.append((capi.METRIC_SATD_X264, orc.orc_satd_x264))
    for metric, f in fns:
        want = np.array([f(n, O.p(A[i]), n, O.p(B[i]), n) for i in range(cnt)], np.uint32)
        np.testing.assert_array_equal(metric_gpu(L, metric, n, A, B), want, err_msg=f"metric {metric} n={n}")
    if n <= 32:
        idx = {4: 0, 8: 1, 16: 2, 32: 3}[n]
        want = np.array([orc.orc_satd_table(idx, O.p(A[i]), n, O.p(B[i]), n) for i in range(cnt)], np.uint32)
        np.testing.assert_array_equal(metric_gpu(L, capi.METRIC_SATD_TAB, n, A, B), want)


def test_block_metrics_empty_and_errors(L):
    assert L.xmi_block_metric_batch_dev(capi.METRIC_SAD, 8, 0, None, None, None, None) == 0
    a = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert L.xmi_block_metric_batch_dev(capi.METRIC_SAD, 12, 1, a.data_ptr(), a.data_ptr(), o.data_ptr(), None) == -1
    assert L.xmi_block_metric_batch_dev(capi.METRIC_SATD_X264, 32, 1, a.data_ptr(), a.data_ptr(), o.data_ptr(), None) == -1
    assert L.xmi_block_metric_batch_dev(99, 8, 1, a.data_ptr(), a.data_ptr(), o.data_ptr(), None) == -1


def tu_chain_gpu(L, n, pix, pred, qp, flags):
    cnt = len(pix)
    dp, dr, dq, df = dev(pix), dev(pred), dev(qp.astype(np.uint8)), dev(flags.astype(np.uint8))
    lev = torch.zeros((cnt, n, n), dtype=torch.int16, device="cuda")
    rec = torch.zeros((cnt, n, n), dtype=torch.uint8, device="cuda")
    sm = torch.zeros(cnt, dtype=torch.int32, device="cuda")
    capi.check(L.xmi_tu_chain_batch_dev(n, cnt, dp.data_ptr(), dr.data_ptr(), dq.data_ptr(), df.data_ptr(),
                                        lev.data_ptr(), rec.data_ptr(), sm.data_ptr(), None))
    torch.cuda.synchronize()
    return lev.cpu().numpy(), rec.cpu().numpy(), sm.cpu().numpy().astype(np.uint32)


@pytest.mark.parametrize("tag,n,dst", [("dct4", 4, 0), ("dst4", 4, 1), ("dct8", 8, 0), ("dct16", 16, 0), ("dct32", 32, 0)])
def test_tu_chain_golden(L, golden_dir, tag, n, dst):
    g = np.load(os.path.join(golden_dir, "transform.npz"))
    qps = g["qps"]
    P, R = g[f"{tag}_pix"], g[f"{tag}_pred"]
    nc = len(P)
    # all (case, qp, slice) combinations in ONE batch with per-TU qp/flags
    pix = np.repeat(P, len(qps) * 2, axis=0)
    pred = np.repeat(R, len(qps) * 2, axis=0)
    qp = np.tile(np.repeat(qps, 2), nc)
    flags = np.tile(np.array([0, 1]), nc * len(qps)) | (2 if dst else 0)
    lev, rec, sm = tu_chain_gpu(L, n, pix, pred, qp, flags)
    want_lev = g[f"{tag}_level"].reshape(-1, n, n)
    want_sum = g[f"{tag}_sum"].reshape(-1)
    want_rec = g[f"{tag}_rec"].reshape(-1, n, n).copy()
    zero = want_sum == 0
    want_rec[zero] = pred[zero]          # cbf = 0: the reference copies the prediction (src/encode.cpp:3014-3018)
    np.testing.assert_array_equal(lev, want_lev)
    np.testing.assert_array_equal(sm, want_sum)
    np.testing.assert_array_equal(rec, want_rec)
    # the stages on their own
    dp, dr, df = dev(P), dev(R), dev(np.full(nc, 2 if dst else 0, np.uint8))
    coef = torch.zeros((nc, n, n), dtype=torch.int16, device="cuda")
    capi.check(L.xmi_sub_dct_batch_dev(n, nc, dp.data_ptr(), dr.data_ptr(), df.data_ptr(), coef.data_ptr(), None))
    np.testing.assert_array_equal(coef.cpu().numpy(), g[f"{tag}_coef"])
    dl = dev(want_lev)
    dq = dev(qp.astype(np.uint8))
    deq = torch.zeros_like(dl)
    capi.check(L.xmi_dequant_batch_dev(n, len(dl), dl.data_ptr(), dq.data_ptr(), deq.data_ptr(), None))
    np.testing.assert_array_equal(deq.cpu().numpy(), g[f"{tag}_deq"].reshape(-1, n, n))
    rc = torch.zeros((len(dl), n, n), dtype=torch.uint8, device="cuda")
    dpr, dfl = dev(pred), dev(flags.astype(np.uint8))
    capi.check(L.xmi_idct_add_batch_dev(n, len(dl), deq.data_ptr(), dpr.data_ptr(), dfl.data_ptr(), rc.data_ptr(), None))
    np.testing.assert_array_equal(rc.cpu().numpy(), g[f"{tag}_rec"].reshape(-1, n, n))


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_quant_dequant_idct_raw_golden(L, golden_dir, n):
    """raw coefficient inputs up to +-32767: exercises int16 saturation in dequant and inverse clipping"""
    g = np.load(os.path.join(golden_dir, "quant_raw.npz"))
    qps = g["qps"]
    Cs = g[f"q{n}_in"]
    nc = len(Cs)
    coef = np.repeat(Cs, len(qps) * 2, axis=0)
    qp = np.tile(np.repeat(qps, 2), nc).astype(np.uint8)
    flags = np.tile(np.array([0, 1], np.uint8), nc * len(qps))
    dc, dq, df = dev(coef), dev(qp), dev(flags)
    lev = torch.zeros_like(dc)
    sm = torch.zeros(len(coef), dtype=torch.int32, device="cuda")
    capi.check(L.xmi_quant_batch_dev(n, len(coef), dc.data_ptr(), dq.data_ptr(), df.data_ptr(), lev.data_ptr(), sm.data_ptr(), None))
    np.testing.assert_array_equal(lev.cpu().numpy(), g[f"q{n}_level"].reshape(-1, n, n))
    np.testing.assert_array_equal(sm.cpu().numpy().astype(np.uint32), g[f"q{n}_sum"].reshape(-1))
    coef1 = np.repeat(Cs, len(qps), axis=0)
    dc1, dq1 = dev(coef1), dev(np.tile(qps, nc).astype(np.uint8))
    deq = torch.zeros_like(dc1)
    capi.check(L.xmi_dequant_batch_dev(n, len(coef1), dc1.data_ptr(), dq1.data_ptr(), deq.data_ptr(), None))
    np.testing.assert_array_equal(deq.cpu().numpy(), g[f"q{n}_deq"].reshape(-1, n, n))
    for dst, tag in (((0, "idct"),) if n != 4 else ((0, "idct"), (2, "idst"))):
        dcs, dpr = dev(Cs), dev(g[f"{tag}{n}_pred"])
        dfl = dev(np.full(nc, dst, np.uint8))
        rc = torch.zeros((nc, n, n), dtype=torch.uint8, device="cuda")
        capi.check(L.xmi_idct_add_batch_dev(n, nc, dcs.data_ptr(), dpr.data_ptr(), dfl.data_ptr(), rc.data_ptr(), None))
        np.testing.assert_array_equal(rc.cpu().numpy(), g[f"{tag}{n}_rec"])


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_tu_chain_replays_encoder_trace(L, golden_dir, n):
    """The chains the reference's encoder ran while coding the 416x240 clip at QP 32 and 22 (luma and chroma, intra and
    inter, DST and DCT; recorded from bit-identical encodes, tests/golden/tu_trace.npz) as one batch: the encoder's own
    levels, level sums and reconstructed blocks come back."""
    g = np.load(os.path.join(golden_dir, "tu_trace.npz"))
    meta = g[f"n{n}_meta"]                      # dst4, qp, slice, poc, stride, sum, has_rec
    pix, pred = g[f"n{n}_pix"], g[f"n{n}_pred"]
    flags = (meta[:, 2] == O.SLICE_I).astype(np.uint8) | (meta[:, 0].astype(np.uint8) << 1)
    lev, rec, sm = tu_chain_gpu(L, n, pix, pred, meta[:, 1], flags)
    np.testing.assert_array_equal(sm, meta[:, 5].astype(np.uint32))
    np.testing.assert_array_equal(lev, g[f"n{n}_level"])
    done = (meta[:, 6] == 1) | (meta[:, 5] == 0)     # reconstructed by the encoder, or all-zero (prediction copied)
    assert done.sum() > len(meta) // 2
    np.testing.assert_array_equal(rec[done], g[f"n{n}_rec"][done])


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_tu_chain_random_vs_oracle(L, orc, n):
    rng = np.random.default_rng(100 + n)
    cnt = 301 if n <= 8 else 67
    pix = rng.integers(0, 256, (cnt, n, n), dtype=np.uint8)
    pred = np.clip(pix.astype(int) + rng.integers(-30, 31, pix.shape), 0, 255).astype(np.uint8)
    pred[::5] = rng.integers(0, 256, pred[::5].shape, dtype=np.uint8)
    pred[3::11] = pix[3::11]                      # zero residual -> cbf 0 path
    qp = rng.integers(0, 52, cnt)
    flags = rng.integers(0, 4 if n == 4 else 2, cnt)
    lev, rec, sm = tu_chain_gpu(L, n, pix, pred, qp, flags)
    for i in range(cnt):
        wl = np.zeros((n, n), np.int16)
        wr = np.zeros((n, n), np.uint8)
        mode = 1 if (flags[i] & 2) else O.MODE_INVALID
        st = O.SLICE_I if (flags[i] & 1) else O.SLICE_P
        ws = orc.orc_tu_chain(O.p(pix[i]), O.p(pred[i]), O.p(wr), O.p(wl), n, n, n, int(qp[i]), mode, st)
        assert sm[i] == ws, f"sum TU {i}"
        np.testing.assert_array_equal(lev[i], wl, err_msg=f"level TU {i}")
        np.testing.assert_array_equal(rec[i], wr, err_msg=f"rec TU {i}")


@pytest.mark.parametrize("n,cnt,style", [(4, 600_037, "mixed"), (4, 600_000, "dst"), (8, 140_005, "p"), (16, 40_003, "p"), (32, 9_001, "p")])
def test_tu_chain_large_batch_matrix_core_vs_valu(L, n, cnt, style):
    """More tiles than resident wavefronts (the persistent loop and its prefetch) plus a ragged tail: the
    matrix-core kernels against the VALU kernel, which a misaligned per-TU array selects and which the
    oracle tests above pin."""
    g = torch.Generator(device="cuda").manual_seed(n + cnt)
    pix = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device="cuda", generator=g)
    pred = (pix.to(torch.int16) + torch.randint(-40, 41, pix.shape, dtype=torch.int16, device="cuda", generator=g)).clamp(0, 255).to(torch.uint8)
    pred[::7] = torch.randint(0, 256, pred[::7].shape, dtype=torch.uint8, device="cuda", generator=g)
    qp_buf = torch.randint(0, 52, (cnt + 16,), dtype=torch.uint8, device="cuda", generator=g)
    fl_buf = torch.randint(0, 2, (cnt + 16,), dtype=torch.uint8, device="cuda", generator=g)
    if style == "mixed":      # DCT and DST TUs interleaved in runs of random length, so tiles are uniform and mixed
        runs = torch.randint(0, 2, ((cnt + 16) // 48 + 1,), dtype=torch.uint8, device="cuda", generator=g).repeat_interleave(48)[:cnt + 16]
        fl_buf |= runs * 2
    elif style == "dst":
        fl_buf |= 2
    outs = []
    for shift in (0, 1):      # 0: aligned -> matrix cores; 1: per-TU arrays start at an odd address -> VALU kernel
        qp = qp_buf[shift:shift + cnt]
        fl = fl_buf[shift:shift + cnt]
        if shift:             # same values at the shifted address
            qp.copy_(qp_buf[:cnt].clone())
            fl.copy_(fl_buf[:cnt].clone())
        lev = torch.zeros((cnt, n, n), dtype=torch.int16, device="cuda")
        rec = torch.zeros((cnt, n, n), dtype=torch.uint8, device="cuda")
        sm = torch.zeros(cnt, dtype=torch.int32, device="cuda")
        capi.check(L.xmi_tu_chain_batch_dev(n, cnt, pix.data_ptr(), pred.data_ptr(), qp.data_ptr(), fl.data_ptr(),
                                            lev.data_ptr(), rec.data_ptr(), sm.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append((lev, rec, sm, qp.clone(), fl.clone()))
    assert torch.equal(outs[0][3], outs[1][3]) and torch.equal(outs[0][4], outs[1][4])
    assert torch.equal(outs[0][0], outs[1][0]), "levels differ"
    assert torch.equal(outs[0][1], outs[1][1]), "reconstruction differs"
    assert torch.equal(outs[0][2], outs[1][2]), "level sums differ"
    assert outs[0][2].any()


@pytest.mark.parametrize("n,cnt", [(4, 600_037), (8, 140_005), (16, 40_003), (32, 9_001)])
def test_tu_stages_large_batch_fast_vs_valu(L, n, cnt):
    """xSubDct / xQuant / xDeQuant / xIDctAdd batches: the matrix-core and streaming kernels (16-byte aligned
    arrays) against the VALU kernel (the int16 array starts 2 bytes off), large ragged batches, mixed DCT/DST for 4x4."""
    g = torch.Generator(device="cuda").manual_seed(7 * n + cnt)
    pix = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device="cuda", generator=g)
    pred = (pix.to(torch.int16) + torch.randint(-60, 61, pix.shape, dtype=torch.int16, device="cuda", generator=g)).clamp(0, 255).to(torch.uint8)
    qp = torch.randint(0, 52, (cnt,), dtype=torch.uint8, device="cuda", generator=g)
    fl = torch.randint(0, 2, (cnt,), dtype=torch.uint8, device="cuda", generator=g)
    if n == 4:
        fl |= torch.randint(0, 2, (cnt // 48 + 1,), dtype=torch.uint8, device="cuda", generator=g).repeat_interleave(48)[:cnt] * 2
    nn = cnt * n * n

    def i16(shift):       # int16 buffer whose data pointer is 16-byte aligned (shift 0) or 2 bytes past that (shift 1)
        buf = torch.zeros(nn + 8, dtype=torch.int16, device="cuda")
        return buf[shift:shift + nn]

    res = []
    for shift in (0, 1):
        coef, lev, deq = i16(shift), i16(shift), i16(shift)
        rec = torch.zeros((cnt, n, n), dtype=torch.uint8, device="cuda")
        sm = torch.zeros(cnt, dtype=torch.int32, device="cuda")
        capi.check(L.xmi_sub_dct_batch_dev(n, cnt, pix.data_ptr(), pred.data_ptr(), fl.data_ptr(), coef.data_ptr(), None))
        capi.check(L.xmi_quant_batch_dev(n, cnt, coef.data_ptr(), qp.data_ptr(), fl.data_ptr(), lev.data_ptr(), sm.data_ptr(), None))
        capi.check(L.xmi_dequant_batch_dev(n, cnt, lev.data_ptr(), qp.data_ptr(), deq.data_ptr(), None))
        capi.check(L.xmi_idct_add_batch_dev(n, cnt, deq.data_ptr(), pred.data_ptr(), fl.data_ptr(), rec.data_ptr(), None))
        torch.cuda.synchronize()
        res.append((coef.clone(), lev.clone(), sm, deq.clone(), rec))
    for name, a, b in zip(("coefficients", "levels", "sums", "dequantised", "reconstruction"), res[0], res[1]):
        assert torch.equal(a, b), name + " differ"
    # and the four stages chained equal the fused chain
    lev = torch.zeros(nn, dtype=torch.int16, device="cuda")
    rec = torch.zeros((cnt, n, n), dtype=torch.uint8, device="cuda")
    sm = torch.zeros(cnt, dtype=torch.int32, device="cuda")
    capi.check(L.xmi_tu_chain_batch_dev(n, cnt, pix.data_ptr(), pred.data_ptr(), qp.data_ptr(), fl.data_ptr(),
                                        lev.data_ptr(), rec.data_ptr(), sm.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(lev, res[0][1]) and torch.equal(sm, res[0][2]) and torch.equal(rec, res[0][4])


def test_tu_roundtrip_property_full_size(L):
    """size-independent property at a large batch: qp 0 on an all-equal block reconstructs it exactly, and
    a zero residual gives zero levels / rec == pred for every TU of the batch"""
    n, cnt = 32, 4096
    rng = np.random.default_rng(9)
    pred = rng.integers(0, 256, (cnt, n, n), dtype=np.uint8)
    lev, rec, sm = tu_chain_gpu(L, n, pred, pred, np.full(cnt, 22), np.zeros(cnt, int))
    assert not lev.any() and not sm.any()
    np.testing.assert_array_equal(rec, pred)
    pix = np.clip(pred.astype(int) + 8, 0, 255).astype(np.uint8)   # DC offset survives qp 0 exactly where unclipped
    lev, rec, sm = tu_chain_gpu(L, n, pix, pred, np.zeros(cnt, int), np.zeros(cnt, int))
    ok = (pred.astype(int) + 8 <= 255).all(axis=(1, 2))
    np.testing.assert_array_equal(rec[ok], pix[ok])
    assert (sm[ok] > 0).all()


def test_percall_dropins_match_oracle(L, orc):
    """host-pointer, strided drop-ins (what the reference's tables would call)"""
    rng = np.random.default_rng(77)
    src = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    ref = rng.integers(0, 256, (80, 96), dtype=np.uint8)
    for n in (4, 8, 16, 32, 64):
        assert L.xmi_sad(n, O.p(src), 64, O.at(ref, 3, 5), 96) == orc.orc_sad(n, O.p(src), 64, O.at(ref, 3, 5), 96)
        assert L.xmi_ssd(n, O.p(src), 64, O.at(ref, 3, 5), 96) == orc.orc_ssd(n, O.p(src), 64, O.at(ref, 3, 5), 96)
        assert L.xmi_satd_hm(n, O.p(src), 64, O.at(ref, 3, 5), 96) == orc.orc_satd_hm(n, O.p(src), 64, O.at(ref, 3, 5), 96)
    for idx, n in enumerate((4, 8, 16, 32)):
        assert L.xmi_satd(n, O.p(src), 64, O.at(ref, 1, 2), 96) == orc.orc_satd_table(idx, O.p(src), 64, O.at(ref, 1, 2), 96)
    assert L.xmi_satd(64, O.p(src), 64, O.p(ref), 96) == -1
    pred = np.clip(src.astype(int) + rng.integers(-20, 21, src.shape), 0, 255).astype(np.uint8)
    for n, mode in ((4, O.MODE_INVALID), (4, 7), (8, O.MODE_INVALID), (16, O.MODE_INVALID), (32, O.MODE_INVALID)):
        for qp, st in ((22, O.SLICE_P), (37, O.SLICE_I), (0, O.SLICE_I), (51, O.SLICE_P)):
            c0 = np.full((64, 64), 1234, np.int16)
            c1 = c0.copy()
            capi.check(L.xmi_sub_dct(O.p(c0), O.p(src), 64, O.p(pred), 64, n, mode))
            orc.orc_sub_dct(O.p(c1), O.p(src), 64, O.p(pred), 64, n, mode)
            np.testing.assert_array_equal(c0, c1)          # also: nothing outside the n x n block was touched
            l0, l1 = np.full((64, 64), -7, np.int16), np.full((64, 64), -7, np.int16)
            s0 = L.xmi_quant(O.p(l0), O.p(c0), 64, qp, n, n, st)
            s1 = orc.orc_quant(O.p(l1), O.p(c1), 64, qp, n, n, st)
            assert s0 == s1
            np.testing.assert_array_equal(l0, l1)
            d0, d1 = np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16)
            capi.check(L.xmi_dequant(O.p(d0), O.p(l0), 64, qp, n, n))
            orc.orc_dequant(O.p(d1), O.p(l1), 64, qp, n, n)
            np.testing.assert_array_equal(d0, d1)
            r0, r1 = np.zeros((64, 80), np.uint8), np.zeros((64, 80), np.uint8)
            capi.check(L.xmi_idct_add(O.p(r0), 80, O.p(d0), 64, O.p(pred), n, mode))
            orc.orc_idct_add(O.p(r1), 80, O.p(d1), 64, O.p(pred), n, mode)
            np.testing.assert_array_equal(r0, r1)
            r2, l2 = np.zeros((64, 80), np.uint8), np.zeros((64, 64), np.int16)
            r3, l3 = np.zeros((64, 80), np.uint8), np.zeros((64, 64), np.int16)
            s2 = L.xmi_tu_chain(O.p(src), O.p(pred), O.p(r2), O.p(l2), n, 64, 80, qp, mode, st)
            s3 = orc.orc_tu_chain(O.p(src), O.p(pred), O.p(r3), O.p(l3), n, 64, 80, qp, mode, st)
            assert s2 == s3
            np.testing.assert_array_equal(l2, l3)
            np.testing.assert_array_equal(r2, r3)
    assert L.xmi_quant(O.p(l0), O.p(c0), 64, 52, 8, 8, O.SLICE_P) == -1
    assert L.xmi_sub_dct(O.p(c0), O.p(src), 64, O.p(pred), 64, 64, 0) == -1


def test_percall_dropins_from_concurrent_threads(L, orc):
    """The reference calls its primitives from up to 16 OpenMP row threads at once (inc/config.h:109). Eight host threads,
    each with its own blocks, interleave the per-call entry points (ctypes releases the interpreter lock during a call):
    every result is the oracle's for that thread's data, and the error text of one thread does not show in another."""
    import threading
    expect, blocks = [], []
    for t in range(8):
        rng = np.random.default_rng(500 + t)
        src = rng.integers(0, 256, (64, 64), dtype=np.uint8)
        ref = rng.integers(0, 256, (72, 96), dtype=np.uint8)
        pred = np.clip(src.astype(int) + rng.integers(-12, 13, src.shape), 0, 255).astype(np.uint8)
        blocks.append((src, ref, pred))
        e = []
        for n in (8, 16, 32, 64):
            e.append(orc.orc_sad(n, O.p(src), 64, O.at(ref, 2, 3), 96))
            e.append(orc.orc_satd_hm(n, O.p(src), 64, O.at(ref, 2, 3), 96))
        for n in (4, 8, 16, 32):
            r, l = np.zeros((64, 80), np.uint8), np.zeros((64, 64), np.int16)
            s = orc.orc_tu_chain(O.p(src), O.p(pred), O.p(r), O.p(l), n, 64, 80, 27 + t, O.MODE_INVALID, O.SLICE_P)
            e.append((s, r.tobytes(), l.tobytes()))
        expect.append(e)
    failures = []

    def work(t):
        src, ref, pred = blocks[t]
        try:
            for rep in range(6):
                got = []
                for n in (8, 16, 32, 64):
                    got.append(L.xmi_sad(n, O.p(src), 64, O.at(ref, 2, 3), 96))
                    got.append(L.xmi_satd_hm(n, O.p(src), 64, O.at(ref, 2, 3), 96))
                for n in (4, 8, 16, 32):
                    r, l = np.zeros((64, 80), np.uint8), np.zeros((64, 64), np.int16)
                    s = L.xmi_tu_chain(O.p(src), O.p(pred), O.p(r), O.p(l), n, 64, 80, 27 + t, O.MODE_INVALID, O.SLICE_P)
                    got.append((s, r.tobytes(), l.tobytes()))
                if got != expect[t]:
                    failures.append(f"thread {t} repetition {rep}: results differ from the oracle")
                if t == 0:          # provoke an error in one thread only
                    assert L.xmi_satd(64, O.p(src), 64, O.p(ref), 96) == -1
                elif rep == 5 and b"xSatdN" in L.xmi_last_error():
                    failures.append(f"thread {t} sees thread 0's error text")
        except Exception as exc:       # noqa: BLE001 - reported below, from the main thread
            failures.append(f"thread {t}: {exc!r}")

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not failures, failures[:3]
