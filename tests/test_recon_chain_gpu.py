"""GPU: from decisions to reconstruction on the device for the luma of a P picture - motion vectors per 32x32 block (as a
search would return them) -> prediction blocks (xPredInterLumaBlk) -> transform chains (xSubDctQuantRec) -> reconstruction
and levels - against the oracle's prediction and chain on the host, block by block."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O
from tests.test_me_search_gpu import make_video
from x265_amd import capi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
PAD = 80


@pytest.mark.parametrize("n,qp", [(32, 32), (16, 27), (8, 22)])
def test_prediction_feeds_the_transform_chain(n, qp):
    L = capi.load()
    capi.check(L.xmi_init(0))
    orc = O.load()
    w, h = 256, 128
    cur, ref = make_video(w, h, 44, (2, 1), 8)
    stride = ref.strides[0]
    rng = np.random.default_rng(45)
    blocks = [(x, y) for y in range(0, h, n) for x in range(0, w, n)]
    cnt = len(blocks)
    reqs = (capi.InterPredRequest * cnt)()
    for k, (x, y) in enumerate(blocks):
        reqs[k].x0, reqs[k].y0, reqs[k].size = x, y, n
        reqs[k].mv[0], reqs[k].mv[1] = -8 + int(rng.integers(-9, 10)), -4 + int(rng.integers(-9, 10))      # around the true motion, all phases
        reqs[k].out_y = k * n * n
    d_ref, d_req = torch.from_numpy(ref).cuda(), torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_pred = torch.zeros(cnt * n * n, dtype=torch.uint8, device="cuda")
    # source blocks in the same packed order (the staging a CTU loop would do once per picture)
    pix = np.ascontiguousarray(cur.reshape(h // n, n, w // n, n).transpose(0, 2, 1, 3)).reshape(cnt, n, n)
    d_pix = torch.from_numpy(pix).cuda()
    d_qp = torch.full((cnt,), qp, dtype=torch.uint8, device="cuda")
    d_flags = torch.zeros(cnt, dtype=torch.uint8, device="cuda")        # P slice, DCT
    d_level = torch.zeros(cnt * n * n, dtype=torch.int16, device="cuda")
    d_rec = torch.zeros(cnt * n * n, dtype=torch.uint8, device="cuda")
    d_sum = torch.zeros(cnt, dtype=torch.int32, device="cuda")
    capi.check(L.xmi_pred_inter_batch_dev(d_ref.data_ptr() + PAD * stride + PAD, stride, None, None, 0, d_req.data_ptr(), cnt, d_pred.data_ptr(), None, None, None))
    capi.check(L.xmi_tu_chain_batch_dev(n, cnt, d_pix.data_ptr(), d_pred.data_ptr(), d_qp.data_ptr(), d_flags.data_ptr(), d_level.data_ptr(), d_rec.data_ptr(),
                                        d_sum.data_ptr(), None))
    torch.cuda.synchronize()
    level, rec, sums = d_level.cpu().numpy().reshape(cnt, n, n), d_rec.cpu().numpy().reshape(cnt, n, n), d_sum.cpu().numpy()
    pred_h, rec_h, lev_h = np.zeros((64, 64), np.uint8), np.zeros((64, 80), np.uint8), np.zeros((64, 64), np.int16)
    src_h = np.zeros((64, 64), np.uint8)
    coded = 0
    for k, (x, y) in enumerate(blocks):
        orc.orc_pred_inter_luma(O.at(ref, PAD + y, PAD + x), O.p(pred_h), stride, n, n, reqs[k].mv[0], reqs[k].mv[1])
        src_h[:n, :n] = cur[y:y + n, x:x + n]
        s = orc.orc_tu_chain(O.p(src_h), O.p(pred_h), O.p(rec_h), O.p(lev_h), n, 64, 80, qp, O.MODE_INVALID, O.SLICE_P)
        assert int(sums[k]) == s, f"block {k}: level sum"
        np.testing.assert_array_equal(level[k], lev_h[:n, :n], err_msg=f"block {k}: levels")
        np.testing.assert_array_equal(rec[k], rec_h[:n, :n], err_msg=f"block {k}: reconstruction")
        coded += s != 0
    assert coded > cnt // 4          # the chain is exercised beyond the all-zero shortcut


@pytest.mark.parametrize("n,qp", [(4, 27), (8, 32), (16, 22), (32, 37)])
def test_intra_prediction_feeds_the_transform_chain(n, qp):
    """The I-slice form of the same step: neighbouring samples and a mode per block -> xEncIntraPred blocks -> chains with
    the intra rounding (and the DST for 4x4 luma) -> levels and reconstruction, against the oracle."""
    L = capi.load()
    capi.check(L.xmi_init(0))
    orc = O.load()
    rng = np.random.default_rng(50 + n)
    cnt = 300
    refs = np.zeros((cnt, capi.INTRA_REF_PITCH), np.uint8)
    base = rng.integers(40, 216, cnt)
    refs[:, :4 * n + 1] = np.clip(base[:, None] + rng.integers(-30, 31, (cnt, 4 * n + 1)), 0, 255)
    modes = rng.integers(0, 35, cnt)
    is_luma = int(n != 32)
    reqs = (capi.IntraPredRequest * cnt)()
    for k in range(cnt):
        reqs[k].size, reqs[k].mode, reqs[k].is_luma, reqs[k].out = n, int(modes[k]), is_luma, k * n * n
    pix = np.clip(base[:, None, None] + rng.integers(-40, 41, (cnt, n, n)), 0, 255).astype(np.uint8)
    d_refs, d_req = torch.from_numpy(refs).cuda(), torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_pred = torch.zeros(cnt * n * n, dtype=torch.uint8, device="cuda")
    d_pix = torch.from_numpy(pix).cuda()
    d_qp = torch.full((cnt,), qp, dtype=torch.uint8, device="cuda")
    d_flags = torch.full((cnt,), 1 | (2 if n == 4 else 0), dtype=torch.uint8, device="cuda")      # I slice; DST for 4x4 luma
    d_level = torch.zeros(cnt * n * n, dtype=torch.int16, device="cuda")
    d_rec = torch.zeros(cnt * n * n, dtype=torch.uint8, device="cuda")
    d_sum = torch.zeros(cnt, dtype=torch.int32, device="cuda")
    capi.check(L.xmi_intra_pred_batch_dev(d_refs.data_ptr(), d_req.data_ptr(), cnt, d_pred.data_ptr(), None))
    capi.check(L.xmi_tu_chain_batch_dev(n, cnt, d_pix.data_ptr(), d_pred.data_ptr(), d_qp.data_ptr(), d_flags.data_ptr(), d_level.data_ptr(), d_rec.data_ptr(),
                                        d_sum.data_ptr(), None))
    torch.cuda.synchronize()
    level, rec, sums = d_level.cpu().numpy().reshape(cnt, n, n), d_rec.cpu().numpy().reshape(cnt, n, n), d_sum.cpu().numpy()
    pred_h, rec_h, lev_h, src_h = np.zeros((64, 64), np.uint8), np.zeros((64, 80), np.uint8), np.zeros((64, 64), np.int16), np.zeros((64, 64), np.uint8)
    coded = 0
    for k in range(cnt):
        rf = np.ascontiguousarray(refs[k])
        orc.orc_intra_pred(O.p(pred_h), O.p(rf), 64, int(modes[k]), n, is_luma)
        src_h[:n, :n] = pix[k]
        s = orc.orc_tu_chain(O.p(src_h), O.p(pred_h), O.p(rec_h), O.p(lev_h), n, 64, 80, qp, int(modes[k]), O.SLICE_I)
        assert int(sums[k]) == s, f"block {k}: level sum"
        np.testing.assert_array_equal(level[k], lev_h[:n, :n], err_msg=f"block {k}: levels")
        np.testing.assert_array_equal(rec[k], rec_h[:n, :n], err_msg=f"block {k}: reconstruction")
        coded += s != 0
    assert coded > cnt // 4
