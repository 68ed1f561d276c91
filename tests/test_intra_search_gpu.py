"""GPU: batched intra luma mode search (xmi_intra_search_batch_dev) against the stored outputs of the reference's
compiled prediction code (tests/golden/intra.npz, 32x32 incl. the decision) and against the oracle's restatement
of src/encode.cpp:4005-4081 on seeded requests of every size. Every output must match exactly (SATD per mode,
best mode, coded index, cost as a double)."""
import ctypes as C
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


def run_gpu(L, frame, reqs, refs):
    n = len(reqs)
    d_src = torch.from_numpy(frame).cuda()
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_ref = torch.from_numpy(refs).cuda()
    d_res = torch.zeros(n * C.sizeof(capi.IntraResult), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_intra_search_batch_dev(d_src.data_ptr(), frame.strides[0], d_ref.data_ptr(), d_req.data_ptr(), n,
                                            d_res.data_ptr(), None))
    torch.cuda.synchronize()
    return (capi.IntraResult * n).from_buffer_copy(d_res.cpu().numpy().tobytes())


def test_intra_search_matches_reference_golden(L):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "intra.npz"))
    orc = O.load()
    for n in (4, 8, 16, 32):
        t = f"n{n}_"
        raw, src = g[t + "raw"], g[t + "src"]
        nc = len(raw)
        frame = np.zeros((32, 32 * nc), np.uint8)
        reqs = (capi.IntraRequest * nc)()
        refs = np.zeros((nc, capi.INTRA_REF_PITCH), np.uint8)
        for ci in range(nc):
            frame[:n, 32 * ci:32 * ci + n] = src[ci]
            rq = reqs[ci]
            rq.x0, rq.y0, rq.size = 32 * ci, 0, n
            rq.valids, rq.strong_smoothing = int(g[t + "valids"][ci]), int(g[t + "strong"][ci])
            rq.mpm[0], rq.mpm[1], rq.mpm[2] = [int(v) for v in g[t + "mpm"][ci]]
            rq.sqrt_lambda = float(g[t + "sqrt_lambda"][ci])
            refs[ci, :4 * n + 1] = raw[ci]
        res = run_gpu(L, frame, reqs, refs)
        for ci in range(nc):
            if n == 32:      # SATD and decision from the reference's own functions
                np.testing.assert_array_equal(np.array(res[ci].satd[:], np.uint32), g[t + "satd"][ci], err_msg=f"satd case {ci}")
                assert (res[ci].best_mode, res[ci].pred_idx) == tuple(int(v) for v in g[t + "best"][ci])
                assert res[ci].best_cost == float(g[t + "cost"][ci])
            else:            # SATD of the reference's predictions (x264 form, restated) per mode
                s = np.ascontiguousarray(src[ci])
                for mode in range(35):
                    pr = np.ascontiguousarray(g[t + "pred"][ci, mode])
                    want = orc.orc_satd_x264(n, O.p(s), n, O.p(pr), n)
                    assert res[ci].satd[mode] == want, f"size {n} case {ci} mode {mode}"


@pytest.mark.parametrize("seed", [1, 2])
def test_intra_search_matches_oracle(L, seed):
    orc = O.load()
    rng = np.random.default_rng(seed)
    H, W, nreq = 128, 256, 400
    yy, xx = np.mgrid[0:H, 0:W]
    frame = np.clip(128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 30 * np.sin((xx + 3 * yy) / 4.0) + rng.integers(-12, 13, (H, W)), 0, 255).astype(np.uint8)
    reqs = (capi.IntraRequest * nreq)()
    refs = np.zeros((nreq, capi.INTRA_REF_PITCH), np.uint8)
    for i in range(nreq):
        n = int(rng.choice([4, 8, 16, 32]))
        rq = reqs[i]
        rq.size = n
        rq.x0 = int(rng.integers(0, (W - n) // 4 + 1)) * 4
        rq.y0 = int(rng.integers(0, (H - n) // 4 + 1)) * 4
        rq.valids = int(rng.choice([0, 0x1f, 0x1e, 0x0e, 0x0c, 0x06, 0x02, 0x08, 0x1a, 0x04, 0x10, 0x01, 0x11]))
        rq.strong_smoothing = int(rng.integers(0, 2))
        m = rng.choice(35, 3, replace=bool(i % 9 == 0))
        rq.mpm[0], rq.mpm[1], rq.mpm[2] = int(m[0]), int(m[1]), int(m[2])
        rq.sqrt_lambda = float(rng.choice([0.0, 2.1, 4.7, 13.25, 40.0, 1e3]))
        style = i % 3
        L4 = 4 * n + 1
        if style == 0:       # neighbours that continue the picture content
            base = frame[rq.y0:rq.y0 + 1, rq.x0:rq.x0 + n].mean()
            r = np.clip(base + rng.integers(-25, 26, L4), 0, 255)
        elif style == 1:     # smooth: strong-filter candidates
            r = np.clip(90 + np.arange(L4) * rng.uniform(0, 0.5) + rng.integers(-1, 2, L4), 0, 255)
        else:
            r = rng.integers(0, 256, L4)
        refs[i, :L4] = r.astype(np.uint8)
    res = run_gpu(L, frame, reqs, refs)
    for i in range(nreq):
        rq = reqs[i]
        out = O.OrcIntraResult()
        mpm = np.array([rq.mpm[0], rq.mpm[1], rq.mpm[2]], np.uint8)
        orc.orc_intra_search(O.at(frame, rq.y0, rq.x0), frame.strides[0], O.p(np.ascontiguousarray(refs[i])), rq.size, rq.valids,
                             rq.strong_smoothing, O.p(mpm), rq.sqrt_lambda, C.byref(out))
        assert list(res[i].satd) == list(out.satd), f"request {i} size {rq.size} valids {rq.valids:#x}"
        assert (res[i].best_mode, res[i].pred_idx, res[i].best_cost) == (out.best_mode, out.pred_idx, out.best_cost), f"request {i}"


def test_intra_search_replays_encoder_trace(L):
    """The kernel on the searches the reference's encoder ran (tests/golden/intra_trace.npz, recorded from a
    bit-identical encode): the 35 SATDs of every stored PU, and the decision that follows from them."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "intra_trace.npz"))
    for n in (4, 8, 16, 32):
        t = f"n{n}_"
        src, raw, want = g[t + "src"], g[t + "raw"], g[t + "satd"]
        nc = len(src)
        per_row = 64
        frame = np.zeros((((nc + per_row - 1) // per_row) * 32, per_row * 32), np.uint8)
        reqs = (capi.IntraRequest * nc)()
        refs = np.zeros((nc, capi.INTRA_REF_PITCH), np.uint8)
        for i in range(nc):
            y0, x0 = (i // per_row) * 32, (i % per_row) * 32
            frame[y0:y0 + n, x0:x0 + n] = src[i]
            rq = reqs[i]
            rq.x0, rq.y0, rq.size = x0, y0, n
            rq.valids, rq.strong_smoothing = int(g[t + "valids"][i]), int(g[t + "strong"][i])
            rq.mpm[0], rq.mpm[1], rq.mpm[2] = 0, 1, 26
            rq.sqrt_lambda = 4.7
            refs[i, :4 * n + 1] = raw[i]
        res = run_gpu(L, frame, reqs, refs)
        for i in range(nc):
            np.testing.assert_array_equal(np.array(res[i].satd[:], np.uint32), want[i], err_msg=f"{n}x{n} search {i}")
            cost = want[i].astype(np.float64) + np.where(np.arange(35) == 0, 1.0, np.where(np.isin(np.arange(35), (1, 26)), 2.0, 3.0)) * 4.7
            assert res[i].best_mode == int(np.argmin(cost)) and res[i].best_cost == cost.min()


def test_neighbour_gather_feeds_the_search(L):
    """xmi_intra_gather_dev on the searches of tests/golden/intra_trace.npz: every PU is placed in a picture whose
    reconstruction holds the samples the reference had around it (raw, as recorded before xPaddingRef) and whose map of
    reconstructed 4x4 units says what the reference's mode cache said (valids). The kernel must return those availability
    bits and those samples for the available segments, 0x80 elsewhere, and the search run on its output must give the
    recorded SATDs. PUs in the picture's corners and on its edges cover neighbours outside the picture."""
    torch = pytest.importorskip("torch")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "intra_trace.npz"))
    for n in (4, 8, 16, 32):
        t = f"n{n}_"
        src, raw, want, valids = g[t + "src"], g[t + "raw"], g[t + "satd"], g[t + "valids"]
        nc, per_row, cell = len(src), 32, 96                       # one PU per 96x96 cell, at (4, 4) inside it
        rows = (nc + per_row - 1) // per_row
        w, h = per_row * cell, rows * cell
        frame = np.zeros((h, w), np.uint8)
        rec = np.full((h, w), 0x11, np.uint8)
        coded = np.zeros((h // 4, w // 4), np.uint8)
        reqs = (capi.IntraRequest * nc)()
        for i in range(nc):
            y0, x0 = (i // per_row) * cell + 4, (i % per_row) * cell + 4
            frame[y0:y0 + n, x0:x0 + n] = src[i]
            rec[y0:y0 + 2 * n, x0 - 1] = raw[i][:2 * n][::-1]
            rec[y0 - 1, x0 - 1:x0 + 2 * n] = raw[i][2 * n:]
            v = int(valids[i])
            for bit, (ux, uy) in enumerate(((x0 - 4, y0 + n), (x0 - 4, y0), (x0 - 4, y0 - 4), (x0, y0 - 4), (x0 + n, y0 - 4))):
                coded[uy // 4, ux // 4] = (v >> bit) & 1
            rq = reqs[i]
            rq.x0, rq.y0, rq.size, rq.valids = x0, y0, n, 0xff
            rq.strong_smoothing = int(g[t + "strong"][i])
            rq.mpm[0], rq.mpm[1], rq.mpm[2] = 0, 1, 26
            rq.sqrt_lambda = 4.7
        d_rec, d_coded = torch.from_numpy(rec).cuda(), torch.from_numpy(coded).cuda()
        d_req = torch.from_numpy(np.frombuffer(reqs, np.uint8).copy()).cuda()
        d_refs = torch.zeros((nc, capi.INTRA_REF_PITCH), dtype=torch.uint8, device="cuda")
        capi.check(L.xmi_intra_gather_dev(d_rec.data_ptr(), w, w, h, d_coded.data_ptr(), w // 4, d_req.data_ptr(), nc, d_refs.data_ptr(), None))
        torch.cuda.synchronize()
        got_req = (capi.IntraRequest * nc).from_buffer_copy(d_req.cpu().numpy().tobytes())
        refs = d_refs.cpu().numpy()
        seg = [(0, n), (n, 2 * n), (2 * n, 2 * n + 1), (2 * n + 1, 3 * n + 1), (3 * n + 1, 4 * n + 1)]
        for i in range(nc):
            assert got_req[i].valids == int(valids[i]), f"{n}x{n} search {i}"
            for bit, (lo, hi) in enumerate(seg):
                expect = raw[i][lo:hi] if (int(valids[i]) >> bit) & 1 else np.full(hi - lo, 0x80, np.uint8)
                np.testing.assert_array_equal(refs[i, lo:hi], expect, err_msg=f"{n}x{n} search {i} segment {bit}")
        res = run_gpu(L, frame, got_req, refs)
        for i in range(nc):
            np.testing.assert_array_equal(np.array(res[i].satd[:], np.uint32), want[i], err_msg=f"{n}x{n} search {i}")
    # neighbours outside the picture: a 64x64 picture, everything reconstructed
    w = h = 64
    rng = np.random.default_rng(5)
    rec = rng.integers(0, 256, (h, w), dtype=np.uint8)
    coded = np.ones((h // 4, w // 4), np.uint8)
    cases = [(0, 0, 32, 0), (32, 0, 32, 0b00011), (0, 32, 32, 0b11000), (32, 32, 32, 0b01110), (16, 16, 16, 0b11111), (48, 48, 16, 0b01110),
             (60, 4, 4, 0b01111), (4, 60, 4, 0b11110)]
    reqs = (capi.IntraRequest * len(cases))()
    for i, (x0, y0, n, _) in enumerate(cases):
        reqs[i].x0, reqs[i].y0, reqs[i].size = x0, y0, n
    d_rec, d_coded = torch.from_numpy(rec).cuda(), torch.from_numpy(coded).cuda()
    d_req = torch.from_numpy(np.frombuffer(reqs, np.uint8).copy()).cuda()
    d_refs = torch.zeros((len(cases), capi.INTRA_REF_PITCH), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_intra_gather_dev(d_rec.data_ptr(), w, w, h, d_coded.data_ptr(), w // 4, d_req.data_ptr(), len(cases), d_refs.data_ptr(), None))
    torch.cuda.synchronize()
    got_req = (capi.IntraRequest * len(cases)).from_buffer_copy(d_req.cpu().numpy().tobytes())
    refs = d_refs.cpu().numpy()
    for i, (x0, y0, n, v) in enumerate(cases):
        assert got_req[i].valids == v, f"case {i}"
        for k in range(4 * n + 1):
            x, y = (x0 - 1, y0 + 2 * n - 1 - k) if k < 2 * n else (x0 - 1 + k - 2 * n, y0 - 1)
            bit = 0 if k < n else 1 if k < 2 * n else 2 if k == 2 * n else 3 if k <= 3 * n else 4
            inside = 0 <= x < w and 0 <= y < h
            assert refs[i, k] == (rec[y, x] if (v >> bit) & 1 and inside else 0x80), f"case {i} sample {k}"


def test_percall_intra_search_equals_batch(L):
    """xmi_intra_search (host pointers, source rows 64 bytes apart as in the encoder's CTU cache) returns what the
    batch entry point returns for the same search."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "intra.npz"))
    for n in (4, 8, 16, 32):
        t = f"n{n}_"
        raw, src = g[t + "raw"], g[t + "src"]
        nc = len(raw)
        frame = np.zeros((32, 32 * nc), np.uint8)
        reqs = (capi.IntraRequest * nc)()
        refs = np.zeros((nc, capi.INTRA_REF_PITCH), np.uint8)
        for ci in range(nc):
            frame[:n, 32 * ci:32 * ci + n] = src[ci]
            rq = reqs[ci]
            rq.x0, rq.y0, rq.size = 32 * ci, 0, n
            rq.valids, rq.strong_smoothing = int(g[t + "valids"][ci]), int(g[t + "strong"][ci])
            rq.mpm[0], rq.mpm[1], rq.mpm[2] = [int(v) for v in g[t + "mpm"][ci]]
            rq.sqrt_lambda = float(g[t + "sqrt_lambda"][ci])
            refs[ci, :4 * n + 1] = raw[ci]
        batch = run_gpu(L, frame, reqs, refs)
        cache = np.zeros((64, 64), np.uint8)
        for ci in range(nc):
            cache[:] = 0
            cache[32:32 + n, 16:16 + n] = src[ci]
            one = capi.IntraResult()
            capi.check(L.xmi_intra_search(O.at(cache, 32, 16), 64, O.p(np.ascontiguousarray(refs[ci])), C.byref(reqs[ci]), C.byref(one)))
            assert bytes(one) == bytes(batch[ci]), f"size {n} case {ci}"
    bad = capi.IntraRequest()
    bad.size = 64
    assert L.xmi_intra_search(O.p(cache), 64, O.p(refs[0]), C.byref(bad), C.byref(one)) == -1


def test_intra_search_empty_and_errors(L):
    assert L.xmi_intra_search_batch_dev(None, 0, None, None, 0, None, None) == 0
    a = torch.zeros(256, dtype=torch.uint8, device="cuda")
    assert L.xmi_intra_search_batch_dev(a.data_ptr(), 6, a.data_ptr(), a.data_ptr(), 1, a.data_ptr(), None) == -1
