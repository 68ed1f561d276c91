"""GPU: the host-pointer frame session (xmi_frame_*), used the way a C/C++ CTU loop would: numpy host arrays in,
numpy host arrays out, no device pointers and no torch on the caller's side.
 * the motion searches the reference's encoder issued for three P pictures (tests/golden/me_trace.npz) replayed
   picture by picture: the session must return the encoder's MVs, costs, SADs and MVP indices;
 * intra searches and TU chains through the session against the oracle."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from x265_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = capi.load()
    capi.check(lib.xmi_init(0))
    return lib


def test_session_replays_encoder_motion_searches(L, golden_dir):
    g = np.load(os.path.join(golden_dir, "me_trace.npz"))
    pad = int(g["pad"][0])
    first = g[f"p{g['pocs'][0]}_cur"]
    h, w = first.shape
    sess = C.c_void_p()
    capi.check(L.xmi_frame_create(w, h, pad, C.byref(sess)))
    try:
        for poc in g["pocs"]:
            cur, ref = np.ascontiguousarray(g[f"p{poc}_cur"]), np.ascontiguousarray(g[f"p{poc}_ref"])
            req, lam, want = g[f"p{poc}_req"], g[f"p{poc}_lambda"], g[f"p{poc}_res"]
            capi.check(L.xmi_frame_begin(sess, O.p(cur), cur.strides[0], O.p(ref), ref.strides[0]))
            n = len(req)
            reqs = (capi.MeRequest * n)()
            for i, r in enumerate(req):
                rq = reqs[i]
                rq.x0, rq.y0, rq.size = int(r[0]), int(r[1]), int(r[2])
                rq.mvp[0][0], rq.mvp[0][1], rq.mvp[1][0], rq.mvp[1][1] = int(r[3]), int(r[4]), int(r[5]), int(r[6])
                rq.zero_mv_tested = int(r[7])
                rq.mv_min_ipel[0], rq.mv_min_ipel[1], rq.mv_max_ipel[0], rq.mv_max_ipel[1] = int(r[8]), int(r[9]), int(r[10]), int(r[11])
                rq.sqrt_lambda = int(math.sqrt(float(lam[i])))
            res = (capi.MeResult * n)()
            capi.check(L.xmi_frame_me_search(sess, reqs, n, 32, res))
            for i in range(n):
                got = (res[i].mv[0], res[i].mv[1], res[i].best_cost, res[i].best_sad, res[i].mvp_idx)
                assert got == tuple(int(v) for v in want[i]), f"picture {poc} request {i}"
            # the same batch with the picture's SAD surface resident: integer candidates become lookups
            assert L.xmi_frame_sad_surface(sess, 128) == -1 and b"pad" in L.xmi_last_error()
            capi.check(L.xmi_frame_sad_surface(sess, 64))
            res2 = (capi.MeResult * n)()
            capi.check(L.xmi_frame_me_search(sess, reqs, n, 32, res2))
            assert bytes(res2) == bytes(res), f"picture {poc}: surface-backed search differs"
        # a request outside the picture is refused, not executed
        bad = (capi.MeRequest * 1)()
        bad[0].x0, bad[0].y0, bad[0].size = w - 4, 0, 8
        assert L.xmi_frame_me_search(sess, bad, 1, 32, (capi.MeResult * 1)()) == -1
        assert b"outside" in L.xmi_last_error()
    finally:
        capi.check(L.xmi_frame_destroy(sess))


def test_session_intra_and_tu_chain_match_oracle(L):
    orc = O.load()
    rng = np.random.default_rng(11)
    H, W, pad = 64, 128, 16
    frame = rng.integers(0, 256, (H, W), dtype=np.uint8)
    refp = rng.integers(0, 256, (H + 2 * pad, W + 2 * pad), dtype=np.uint8)
    sess = C.c_void_p()
    capi.check(L.xmi_frame_create(W, H, pad, C.byref(sess)))
    try:
        capi.check(L.xmi_frame_begin(sess, O.p(frame), W, O.p(refp), refp.strides[0]))
        n = 60
        reqs = (capi.IntraRequest * n)()
        refs = np.zeros((n, capi.INTRA_REF_PITCH), np.uint8)
        for i in range(n):
            size = int(rng.choice([4, 8, 16, 32]))
            rq = reqs[i]
            rq.size, rq.valids, rq.strong_smoothing = size, int(rng.choice([0, 0x1f, 0x0e, 0x18, 0x02])), int(i & 1)
            rq.x0, rq.y0 = int(rng.integers(0, (W - size) // 4 + 1)) * 4, int(rng.integers(0, (H - size) // 4 + 1)) * 4
            rq.mpm[0], rq.mpm[1], rq.mpm[2] = [int(v) for v in rng.choice(35, 3, replace=False)]
            rq.sqrt_lambda = float(rng.choice([2.1, 13.25]))
            refs[i, :4 * size + 1] = rng.integers(0, 256, 4 * size + 1)
        res = (capi.IntraResult * n)()
        capi.check(L.xmi_frame_intra_search(sess, O.p(refs), reqs, n, res))
        for i in range(n):
            rq = reqs[i]
            out = O.OrcIntraResult()
            mpm = np.array(list(rq.mpm), np.uint8)
            orc.orc_intra_search(O.at(frame, rq.y0, rq.x0), W, O.p(np.ascontiguousarray(refs[i])), rq.size, rq.valids, rq.strong_smoothing,
                                 O.p(mpm), rq.sqrt_lambda, C.byref(out))
            assert list(res[i].satd) == list(out.satd) and (res[i].best_mode, res[i].pred_idx, res[i].best_cost) == (out.best_mode, out.pred_idx, out.best_cost)
        for tn, cnt in ((4, 150), (16, 21)):
            pix = rng.integers(0, 256, (cnt, tn, tn), dtype=np.uint8)
            pred = np.clip(pix.astype(int) + rng.integers(-25, 26, pix.shape), 0, 255).astype(np.uint8)
            qp = rng.integers(0, 52, cnt).astype(np.uint8)
            fl = rng.integers(0, 4 if tn == 4 else 2, cnt).astype(np.uint8)
            lev, rec, sm = np.zeros((cnt, tn, tn), np.int16), np.zeros((cnt, tn, tn), np.uint8), np.zeros(cnt, np.uint32)
            capi.check(L.xmi_frame_tu_chain(sess, tn, cnt, O.p(pix), O.p(pred), O.p(qp), O.p(fl), O.p(lev), O.p(rec), O.p(sm)))
            for i in range(cnt):
                wl, wr = np.zeros((tn, tn), np.int16), np.zeros((tn, tn), np.uint8)
                ws = orc.orc_tu_chain(O.p(pix[i]), O.p(pred[i]), O.p(wr), O.p(wl), tn, tn, tn, int(qp[i]), 1 if fl[i] & 2 else O.MODE_INVALID,
                                      O.SLICE_I if fl[i] & 1 else O.SLICE_P)
                assert sm[i] == ws and np.array_equal(lev[i], wl) and np.array_equal(rec[i], wr), f"{tn}x{tn} TU {i}"
    finally:
        capi.check(L.xmi_frame_destroy(sess))
