"""GPU: the motion data of a P picture derived diagonal by diagonal on the device - candidates from the maps (xFillMvpCand),
request table, searches (xMotionSearch), vectors stored back - against the same chain of oracle functions run CTU by CTU
in raster order on the host. Both orders respect what a PU may see (left, above, above-right coded; below-left not), so
every predictor pair, every search result and the final motion field must agree."""
import ctypes as C
import math

import numpy as np
import pytest

from tests import oracle_lib as O
from tests.test_me_search_gpu import make_video
from x265_amd import capi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
PAD = 80


@pytest.mark.parametrize("size", [64, 32])
def test_wavefront_motion_estimation(size):
    L = capi.load()
    capi.check(L.xmi_init(0))
    orc = O.load()
    w, h = 448, 256
    cur, ref = make_video(w, h, 31, (5, -3), 4)
    rng = np.random.default_rng(32)                          # a second motion in the right part of the picture, and a still band
    cur = cur.copy()
    cur[:, 256:] = np.clip(ref[PAD + 4:PAD + 4 + h, PAD - 2 + 256:PAD - 2 + w].astype(int) + rng.integers(-4, 5, (h, w - 256)), 0, 255)
    cur[192:, :192] = ref[PAD + 192:PAD + h, PAD:PAD + 192]
    rows, stride = ref.shape
    lam = 0.578 * 2 ** ((32 - 12) / 3.0)
    d_ref, d_cur = torch.from_numpy(ref).cuda(), torch.from_numpy(cur).cuda()
    planes = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_interp_planes_dev(d_ref.data_ptr(), stride, rows, stride, planes.data_ptr(), None))
    cw, ch = w // 64, h // 64
    per = 64 // size                                        # PUs per CTU side
    d_valid = torch.zeros((h // 4, w // 4), dtype=torch.uint8, device="cuda")
    d_mv = torch.zeros((h // 8, w // 8, 2), dtype=torch.int16, device="cuda")
    got = {}
    # diagonals of CTUs (x + 2 y = d); inside a CTU the PUs in coding order, one launch set per position so that a PU sees its
    # siblings' vectors
    for d in range(cw + 2 * (ch - 1)):
        ctus = [(d - 2 * y, y) for y in range(ch) if 0 <= d - 2 * y < cw]
        for sub in range(per * per):
            sy, sx = (sub >> 1, sub & 1) if per == 2 else (0, 0)
            pus = (capi.MvpRequest * len(ctus))()
            for k, (cx, cy) in enumerate(ctus):
                pus[k].x0, pus[k].y0, pus[k].size_x, pus[k].size_y = cx * 64 + sx * size, cy * 64 + sy * size, size, size
            n = len(ctus)
            d_pu = torch.frombuffer(bytearray(bytes(pus)), dtype=torch.uint8).cuda()
            d_cand = torch.zeros(n * C.sizeof(capi.MvpResult), dtype=torch.uint8, device="cuda")
            d_req = torch.zeros(n * C.sizeof(capi.MeRequest), dtype=torch.uint8, device="cuda")
            d_res = torch.zeros(n * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
            capi.check(L.xmi_mvp_candidates_dev(w, h, d_valid.data_ptr(), d_mv.data_ptr(), d_pu.data_ptr(), n, d_cand.data_ptr(), None))
            capi.check(L.xmi_me_requests_from_candidates_dev(w, h, d_pu.data_ptr(), d_cand.data_ptr(), None, n, int(math.sqrt(lam)), d_req.data_ptr(), None))
            capi.check(L.xmi_me_search_batch_dev(d_cur.data_ptr(), cur.strides[0], planes.data_ptr(), stride, rows, PAD, d_req.data_ptr(), n, 32,
                                                 d_res.data_ptr(), None))
            capi.check(L.xmi_mv_field_store_dev(w, h, d_pu.data_ptr(), d_res.data_ptr(), n, d_valid.data_ptr(), d_mv.data_ptr(), None))
            torch.cuda.synchronize()
            cand = (capi.MvpResult * n).from_buffer_copy(d_cand.cpu().numpy().tobytes())
            res = (capi.MeResult * n).from_buffer_copy(d_res.cpu().numpy().tobytes())
            for k in range(n):
                got[(pus[k].x0, pus[k].y0)] = ((cand[k].amvp[0][0], cand[k].amvp[0][1], cand[k].amvp[1][0], cand[k].amvp[1][1]),
                                               (res[k].mv[0], res[k].mv[1], res[k].best_cost, res[k].best_sad, res[k].mvp_idx))
    # the host: raster order over CTUs, coding order inside
    valid = np.zeros((h // 4, w // 4), np.uint8)
    mv = np.zeros((h // 8, w // 8, 2), np.int16)
    blk = np.zeros((64, 64), np.uint8)
    moved = 0
    for cy in range(ch):
        for cx in range(cw):
            for sub in range(per * per):
                sy, sx = (sub >> 1, sub & 1) if per == 2 else (0, 0)
                x0, y0 = cx * 64 + sx * size, cy * 64 + sy * size
                out = np.zeros((7, 2), np.int16)
                orc.orc_fill_mvp_cand(w, h, O.p(valid), O.p(mv), x0, y0, size, size, 0, 0, O.p(out))
                mn, mx = (C.c_int16 * 2)(), (C.c_int16 * 2)()
                capi.check(L.xmi_cu_mv_clip(cx, cy, cw, ch, mn, mx))
                blk[:] = 0
                blk[:size, :size] = cur[y0:y0 + size, x0:x0 + size]
                mvp = (O.OrcMV * 2)(O.OrcMV(int(out[0, 0]), int(out[0, 1])), O.OrcMV(int(out[1, 0]), int(out[1, 1])))
                r = O.OrcMEResult()
                orc.orc_motion_search(O.p(blk), O.at(ref, PAD + y0, PAD + x0), ref.strides[0], size, mvp, 0, O.OrcMV(mn[0], mn[1]), O.OrcMV(mx[0], mx[1]),
                                      float(lam), 32, C.byref(r))
                want = ((int(out[0, 0]), int(out[0, 1]), int(out[1, 0]), int(out[1, 1])), (r.mv.x, r.mv.y, r.best_cost, r.best_sad, r.mvp_idx))
                assert got[(x0, y0)] == want, f"PU at {x0},{y0}: device {got[(x0, y0)]} host {want}"
                mv[y0 // 8:(y0 + size) // 8, x0 // 8:(x0 + size) // 8] = (r.mv.x, r.mv.y)
                valid[y0 // 4:(y0 + size) // 4, x0 // 4:(x0 + size) // 4] = 1
                moved += (r.mv.x, r.mv.y) != (0, 0)
    np.testing.assert_array_equal(d_mv.cpu().numpy(), mv)
    np.testing.assert_array_equal(d_valid.cpu().numpy(), valid)
    assert moved > len(got) // 2 and len({v[0] for v in got.values()}) >= 4        # real motion, and predictors that differ between PUs
