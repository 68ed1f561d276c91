"""GPU: MV predictor and merge candidate lists (xFillMvpCand) through the C ABI against the reference's compiled function
(tests/golden/mvp_cand.npz) and against the oracle on a larger seeded picture."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from x265_amd import capi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def L():
    lib = capi.load()
    capi.check(lib.xmi_init(0))
    return lib


def run(L, w, h, valid, mv, reqs):
    n = len(reqs)
    d_valid, d_mv = torch.from_numpy(valid).cuda(), torch.from_numpy(mv).cuda()
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_res = torch.full((n * C.sizeof(capi.MvpResult),), 0x5a, dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_mvp_candidates_dev(w, h, d_valid.data_ptr(), d_mv.data_ptr(), d_req.data_ptr(), n, d_res.data_ptr(), None))
    torch.cuda.synchronize()
    return (capi.MvpResult * n).from_buffer_copy(d_res.cpu().numpy().tobytes())


def as_rows(r):
    return [[r.amvp[0][0], r.amvp[0][1]], [r.amvp[1][0], r.amvp[1][1]]] + [[r.merge[k][0], r.merge[k][1]] for k in range(5)]


def test_mvp_candidates_match_reference_golden(L):
    g = np.load(os.path.join(GOLDEN, "mvp_cand.npz"))
    w, h = (int(v) for v in g["size"])
    req = g["req"]
    reqs = (capi.MvpRequest * len(req))()
    for k, r in enumerate(req):
        reqs[k].x0, reqs[k].y0, reqs[k].size_x, reqs[k].size_y, reqs[k].pu_mode, reqs[k].pu_idx = (int(v) for v in r)
    res = run(L, w, h, np.ascontiguousarray(g["valid"]), np.ascontiguousarray(g["mv"]), reqs)
    for k in range(len(req)):
        assert res[k].merge_count == int(g["count"][k]), f"PU {k}: merge count"
        assert as_rows(res[k]) == g["cand"][k].tolist(), f"PU {k}"


def test_mvp_candidates_match_oracle_and_mark_bad_requests(L):
    orc = O.load()
    rng = np.random.default_rng(12)
    w, h = 1920, 1088
    valid = (rng.random((h // 4, w // 4)) < 0.7).astype(np.uint8)
    palette = rng.integers(-64, 65, (6, 2)).astype(np.int16)
    palette[0] = 0
    mv = np.ascontiguousarray(palette[rng.integers(0, 6, (h // 8, w // 8))])
    n = 3000
    reqs = (capi.MvpRequest * n)()
    for k in range(n):
        s = int(rng.choice([8, 16, 32, 64]))
        mode = int(rng.integers(0, 3)) if s > 8 else 0
        idx = int(rng.integers(0, 2)) if mode else 0
        sx, sy = (s, s) if mode == 0 else ((s, s // 2) if mode == 1 else (s // 2, s))
        reqs[k].size_x, reqs[k].size_y, reqs[k].pu_mode, reqs[k].pu_idx = sx, sy, mode, idx
        reqs[k].x0 = int(rng.integers(0, (w - s) // 8 + 1)) * 8 + (sx if (mode == 2 and idx) else 0)
        reqs[k].y0 = int(rng.integers(0, (h - s) // 8 + 1)) * 8 + (sy if (mode == 1 and idx) else 0)
    reqs[5].x0, reqs[6].size_x, reqs[7].y0 = -8, 4, h
    res = run(L, w, h, valid, mv, reqs)
    for k in range(n):
        if k in (5, 6, 7):
            assert res[k].merge_count == -1
            continue
        out = np.zeros((7, 2), np.int16)
        cnt = orc.orc_fill_mvp_cand(w, h, O.p(valid), O.p(mv), reqs[k].x0, reqs[k].y0, reqs[k].size_x, reqs[k].size_y, reqs[k].pu_mode,
                                    reqs[k].pu_idx, O.p(out))
        assert res[k].merge_count == cnt and as_rows(res[k]) == out.tolist(), f"PU {k}"
    assert L.xmi_mvp_candidates_dev(w, h, None, None, None, 0, None, None) == 0
    assert L.xmi_mvp_candidates_dev(w + 4, h, None, None, None, 1, None, None) == -1


def test_percall_form_on_cache_neighbourhoods(L):
    """xmi_fill_mvp_cand with the reference's arguments: the cache neighbourhoods of the golden PUs, rebuilt as
    oracle/gen_golden.py cut them; entries past the count stay as they were."""
    g = np.load(os.path.join(GOLDEN, "mvp_cand.npz"))
    w, h = (int(v) for v in g["size"])
    valid, mv = g["valid"], g["mv"]
    for k in range(0, len(g["req"]), 3):
        x0, y0, sx, sy, mode, idx = (int(v) for v in g["req"][k])
        modes = np.full((18, 18), 255, np.uint8)
        mvs = np.zeros((10, 10, 2), np.int16)
        for r in range(-1, 17):
            for c in range(-1, 17):
                yy, xx = y0 // 4 + r, x0 // 4 + c
                if 0 <= yy < h // 4 and 0 <= xx < w // 4 and valid[yy, xx]:
                    modes[r + 1, c + 1] = 0
        for r in range(-1, 9):
            for c in range(-1, 9):
                yy, xx = y0 // 8 + r, x0 // 8 + c
                if 0 <= yy < h // 8 and 0 <= xx < w // 8:
                    mvs[r + 1, c + 1] = mv[yy, xx]
        out = np.full((7, 2), 0x7777, np.int16)
        cnt = L.xmi_fill_mvp_cand(O.p(out), O.at(modes, 1, 1), C.c_void_p(mvs.ctypes.data + 11 * 4), sx, sy, mode, idx)
        assert cnt == int(g["count"][k]), f"PU {k}"
        np.testing.assert_array_equal(out[:2 + cnt], g["cand"][k][:2 + cnt], err_msg=f"PU {k}")
        assert (out[2 + cnt:] == 0x7777).all()
    assert L.xmi_fill_mvp_cand(O.p(out), O.at(modes, 1, 1), C.c_void_p(mvs.ctypes.data + 11 * 4), 4, 8, 0, 0) == -1


def test_search_requests_built_on_the_device(L):
    """maps -> xmi_mvp_candidates_dev -> xmi_me_requests_from_candidates_dev: every request equals the one put together on the
    host from the candidate kernel's result and xmi_cu_mv_clip; refused or non-square PUs become size-0 requests."""
    rng = np.random.default_rng(5)
    w, h = 448, 256
    valid = (rng.random((h // 4, w // 4)) < 0.75).astype(np.uint8)
    mv = rng.integers(-80, 81, (h // 8, w // 8, 2)).astype(np.int16)
    n = 500
    pus = (capi.MvpRequest * n)()
    for k in range(n):
        s = int(rng.choice([8, 16, 32, 64]))
        pus[k].size_x = pus[k].size_y = s
        pus[k].x0, pus[k].y0 = int(rng.integers(0, (w - s) // 8 + 1)) * 8, int(rng.integers(0, (h - s) // 8 + 1)) * 8
    pus[3].size_y, pus[3].pu_mode = pus[3].size_x // 2 if pus[3].size_x > 8 else 8, 1
    pus[4].x0 = w            # outside: refused by the candidate kernel
    zero = rng.integers(0, 2, n).astype(np.uint8)
    d_valid, d_mv, d_zero = torch.from_numpy(valid).cuda(), torch.from_numpy(mv).cuda(), torch.from_numpy(zero).cuda()
    d_pu = torch.frombuffer(bytearray(bytes(pus)), dtype=torch.uint8).cuda()
    d_cand = torch.zeros(n * C.sizeof(capi.MvpResult), dtype=torch.uint8, device="cuda")
    d_req = torch.full((n * C.sizeof(capi.MeRequest),), 0x5a, dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_mvp_candidates_dev(w, h, d_valid.data_ptr(), d_mv.data_ptr(), d_pu.data_ptr(), n, d_cand.data_ptr(), None))
    capi.check(L.xmi_me_requests_from_candidates_dev(w, h, d_pu.data_ptr(), d_cand.data_ptr(), d_zero.data_ptr(), n, 9, d_req.data_ptr(), None))
    torch.cuda.synchronize()
    cand = (capi.MvpResult * n).from_buffer_copy(d_cand.cpu().numpy().tobytes())
    req = (capi.MeRequest * n).from_buffer_copy(d_req.cpu().numpy().tobytes())
    for k in range(n):
        r = req[k]
        if k == 4 or pus[k].size_x != pus[k].size_y:
            assert r.size == 0, f"PU {k}"
            continue
        mn, mx = (C.c_int16 * 2)(), (C.c_int16 * 2)()
        capi.check(L.xmi_cu_mv_clip(pus[k].x0 // 64, pus[k].y0 // 64, w // 64, h // 64, mn, mx))
        got = (r.x0, r.y0, r.size, r.mvp[0][0], r.mvp[0][1], r.mvp[1][0], r.mvp[1][1], r.zero_mv_tested, r.mv_min_ipel[0], r.mv_min_ipel[1],
               r.mv_max_ipel[0], r.mv_max_ipel[1], r.sqrt_lambda)
        want = (pus[k].x0, pus[k].y0, pus[k].size_x, cand[k].amvp[0][0], cand[k].amvp[0][1], cand[k].amvp[1][0], cand[k].amvp[1][1], int(zero[k]),
                mn[0], mn[1], mx[0], mx[1], 9)
        assert got == want, f"PU {k}"
    assert L.xmi_me_requests_from_candidates_dev(100, h, d_pu.data_ptr(), d_cand.data_ptr(), None, n, 9, d_req.data_ptr(), None) == -1


def test_vectors_stored_back_into_the_maps(L):
    """xmi_mv_field_store_dev: the PU's 8x8 vector entries and 4x4 coded flags, nothing else; refused results and PUs off
    the picture are skipped."""
    rng = np.random.default_rng(6)
    w, h = 256, 128
    pus, taken = [], np.zeros((h // 8, w // 8), bool)
    while len(pus) < 60:                                   # non-overlapping PUs
        s = int(rng.choice([8, 16, 32, 64]))
        x0, y0 = int(rng.integers(0, (w - s) // 8 + 1)) * 8, int(rng.integers(0, (h - s) // 8 + 1)) * 8
        if taken[y0 // 8:(y0 + s) // 8, x0 // 8:(x0 + s) // 8].any():
            continue
        taken[y0 // 8:(y0 + s) // 8, x0 // 8:(x0 + s) // 8] = True
        pus.append((x0, y0, s))
    n = len(pus) + 1
    req = (capi.MvpRequest * n)()
    res = (capi.MeResult * n)()
    want_mv, want_valid = np.full((h // 8, w // 8, 2), 77, np.int16), np.zeros((h // 4, w // 4), np.uint8)
    for k, (x0, y0, s) in enumerate(pus):
        req[k].x0, req[k].y0, req[k].size_x, req[k].size_y = x0, y0, s, s
        res[k].mv[0], res[k].mv[1] = int(rng.integers(-200, 201)), int(rng.integers(-200, 201))
        res[k].mvp_idx = -1 if k % 13 == 5 else int(rng.integers(0, 2))
        if res[k].mvp_idx >= 0:
            want_mv[y0 // 8:(y0 + s) // 8, x0 // 8:(x0 + s) // 8] = (res[k].mv[0], res[k].mv[1])
            want_valid[y0 // 4:(y0 + s) // 4, x0 // 4:(x0 + s) // 4] = 1
    req[n - 1].x0, req[n - 1].y0, req[n - 1].size_x, req[n - 1].size_y = w - 8, 0, 16, 16      # sticks out: skipped
    d_mv = torch.full((h // 8, w // 8, 2), 77, dtype=torch.int16, device="cuda")
    d_valid = torch.zeros((h // 4, w // 4), dtype=torch.uint8, device="cuda")
    d_req = torch.frombuffer(bytearray(bytes(req)), dtype=torch.uint8).cuda()
    d_res = torch.frombuffer(bytearray(bytes(res)), dtype=torch.uint8).cuda()
    capi.check(L.xmi_mv_field_store_dev(w, h, d_req.data_ptr(), d_res.data_ptr(), n, d_valid.data_ptr(), d_mv.data_ptr(), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_mv.cpu().numpy(), want_mv)
    np.testing.assert_array_equal(d_valid.cpu().numpy(), want_valid)
