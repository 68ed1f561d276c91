"""GPU: batched xMotionSearch replay (xmi_me_search_batch_dev) against the oracle's restatement of
src/encode.cpp:2815-3003 on synthetic moving video. Every out-parameter must match exactly."""
import ctypes as C
import math

import numpy as np
import pytest

from tests import oracle_lib as O
from x265_amd import capi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
PAD = 80


@pytest.fixture(scope="module")
def L():
    lib = capi.load()
    capi.check(lib.xmi_init(0))
    return lib


def make_video(w, h, seed, motion=(3, 2), noise=6):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h + 2 * PAD, 0:w + 2 * PAD]
    base = 128 + 60 * np.sin(xx / 17.0) + 40 * np.cos(yy / 11.0) + 25 * np.sin((xx + 2 * yy) / 5.0)
    ref = np.clip(base + rng.integers(-noise, noise + 1, base.shape), 0, 255).astype(np.uint8)
    dx, dy = motion
    cur = np.clip(ref[PAD + dy:PAD + dy + h, PAD + dx:PAD + dx + w].astype(int) + rng.integers(-noise, noise + 1, (h, w)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(cur), np.ascontiguousarray(ref)


def make_surface(L, d_cur, d_ref, w, h, stride, pad, rng_):
    """SAD surface of the pair (xmi_sad_surface_dev) -> (geom, planes)."""
    from x265_amd import layout
    nctu = (w // 64) * (h // 64)
    geom = capi.SadGeom(w, h, w, stride, rng_)
    planes = [torch.zeros(layout.plane_elems(nctu, rng_, lv), dtype=torch.int16 if lv < 2 else torch.int32, device="cuda") for lv in range(4)]
    capi.check(L.xmi_sad_surface_dev(C.byref(geom), d_cur.data_ptr(), d_ref.data_ptr() + pad * stride + pad, planes[0].data_ptr(),
                                     planes[1].data_ptr(), planes[2].data_ptr(), planes[3].data_ptr(), 0, nctu, None))
    return geom, planes


def search(L, d_cur, cur_stride, planes, stride, rows, pad, d_req, nreq, d_res, surface=None):
    if surface is None:
        capi.check(L.xmi_me_search_batch_dev(d_cur.data_ptr(), cur_stride, planes.data_ptr(), stride, rows, pad,
                                             d_req.data_ptr(), nreq, 32, d_res.data_ptr(), None))
    else:
        geom, sp = surface
        capi.check(L.xmi_me_search_surface_batch_dev(d_cur.data_ptr(), cur_stride, planes.data_ptr(), stride, rows, pad, C.byref(geom),
                                                     sp[0].data_ptr(), sp[1].data_ptr(), sp[2].data_ptr(), sp[3].data_ptr(),
                                                     d_req.data_ptr(), nreq, 32, d_res.data_ptr(), None))
    torch.cuda.synchronize()


def run_case(L, orc, w, h, seed, nreq, motion, lambdas, noise=6, surface_range=None):
    cur, ref = make_video(w, h, seed, motion, noise)
    rows, stride = ref.shape
    d_ref = torch.from_numpy(ref).cuda()
    planes = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_interp_planes_dev(d_ref.data_ptr(), stride, rows, stride, planes.data_ptr(), None))
    d_cur = torch.from_numpy(cur).cuda()
    surface = make_surface(L, d_cur, d_ref, w, h, stride, PAD, surface_range) if surface_range else None
    rng = np.random.default_rng(seed + 1)
    reqs = (capi.MeRequest * nreq)()
    for i in range(nreq):
        n = int(rng.choice([8, 16, 32, 64]))
        rq = reqs[i]
        rq.size = n
        rq.x0 = int(rng.integers(0, (w - n) // 8 + 1)) * 8
        rq.y0 = int(rng.integers(0, (h - n) // 8 + 1)) * 8
        style = i % 4
        near = (-4 * motion[0] + int(rng.integers(-6, 7)), -4 * motion[1] + int(rng.integers(-6, 7)))
        if style == 0:      # both predictors near the true motion
            m0, m1 = near, (near[0] + int(rng.integers(-3, 4)), near[1] + int(rng.integers(-3, 4)))
        elif style == 1:    # far apart: two MVP passes
            m0, m1 = (int(rng.integers(-200, 201)), int(rng.integers(-200, 201))), near
        elif style == 2:    # zero predictors
            m0, m1 = (0, 0), (0, 0)
        else:               # anything
            m0, m1 = (int(rng.integers(-256, 257)), int(rng.integers(-256, 257))), (int(rng.integers(-256, 257)), int(rng.integers(-256, 257)))
        rq.mvp[0][0], rq.mvp[0][1], rq.mvp[1][0], rq.mvp[1][1] = m0[0], m0[1], m1[0], m1[1]
        rq.zero_mv_tested = int(rng.integers(0, 2))
        lim = 64 if i % 7 else int(rng.integers(2, 40))      # sometimes a tight clip box
        rq.mv_min_ipel[0], rq.mv_min_ipel[1] = -lim, -(64 if i % 5 else lim // 2)
        rq.mv_max_ipel[0], rq.mv_max_ipel[1] = lim, 64 if i % 3 else lim
        rq.sqrt_lambda = int(math.sqrt(lambdas[i % len(lambdas)]))
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_res = torch.zeros(nreq * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
    search(L, d_cur, cur.strides[0], planes, stride, rows, PAD, d_req, nreq, d_res, surface)
    res = (capi.MeResult * nreq).from_buffer_copy(d_res.cpu().numpy().tobytes())
    blk = np.zeros((64, 64), np.uint8)
    moved = 0
    for i in range(nreq):
        rq = reqs[i]
        n = rq.size
        blk[:] = 0
        blk[:n, :n] = cur[rq.y0:rq.y0 + n, rq.x0:rq.x0 + n]
        mvp = (O.OrcMV * 2)(O.OrcMV(rq.mvp[0][0], rq.mvp[0][1]), O.OrcMV(rq.mvp[1][0], rq.mvp[1][1]))
        out = O.OrcMEResult()
        orc.orc_motion_search(O.p(blk), O.at(ref, PAD + rq.y0, PAD + rq.x0), ref.strides[0], n, mvp, rq.zero_mv_tested,
                              O.OrcMV(rq.mv_min_ipel[0], rq.mv_min_ipel[1]), O.OrcMV(rq.mv_max_ipel[0], rq.mv_max_ipel[1]),
                              float(lambdas[i % len(lambdas)]), 32, C.byref(out))
        got = (res[i].mv[0], res[i].mv[1], res[i].best_cost, res[i].best_sad, res[i].mvp_idx, res[i].n_int_sads, res[i].n_sub_sads)
        want = (out.mv.x, out.mv.y, out.best_cost, out.best_sad, out.mvp_idx, out.n_int_sads, out.n_sub_sads)
        assert got == want, f"request {i} (size {n} at {rq.x0},{rq.y0}): gpu {got} oracle {want}"
        moved += out.n_int_sads > 6
    return moved


def test_me_search_matches_oracle(L):
    orc = O.load()
    # lambda of P slices: 0.578 * 2^((QP-12)/3) for QP 22/27/32/37 (src/encode.cpp:5406)
    lambdas = [0.578 * 2 ** ((qp - 12) / 3.0) for qp in (22, 27, 32, 37)] + [400.0, 2500.0]
    moved = run_case(L, orc, 256, 192, seed=5, nreq=240, motion=(3, 2), lambdas=lambdas)
    assert moved > 20       # the diamond walk is exercised, not only the early exits
    run_case(L, orc, 192, 128, seed=9, nreq=120, motion=(-7, 5), lambdas=lambdas, noise=2)
    run_case(L, orc, 128, 128, seed=2, nreq=60, motion=(0, 0), lambdas=[1.0, 50.0], noise=0)   # exact match: cost < 800 exits


@pytest.mark.parametrize("surface_range", [16, 32, 64])
def test_me_search_on_sad_surface_matches_oracle(L, surface_range):
    """Integer candidates looked up in the resident SAD surface instead of computed: same results. Small ranges put many
    candidates outside the surface window, unaligned PUs are off the CTU quad-tree: both fall back to computing."""
    orc = O.load()
    lambdas = [0.578 * 2 ** ((qp - 12) / 3.0) for qp in (22, 27, 32, 37)] + [400.0, 2500.0]
    moved = run_case(L, orc, 256, 192, seed=5, nreq=240, motion=(3, 2), lambdas=lambdas, surface_range=surface_range)
    assert moved > 20
    run_case(L, orc, 192, 128, seed=9, nreq=120, motion=(-7, 5), lambdas=lambdas, noise=2, surface_range=surface_range)
    run_case(L, orc, 128, 128, seed=2, nreq=60, motion=(0, 0), lambdas=[1.0, 50.0], noise=0, surface_range=surface_range)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_me_search_matches_reference_golden(L, tag):
    """The kernel against the stored outputs of the reference's compiled xMotionSearch (tests/golden/me_search.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "me_search.npz"))
    cur, ref, req, want = np.ascontiguousarray(g[f"{tag}_cur"]), np.ascontiguousarray(g[f"{tag}_ref"]), g[f"{tag}_req"], g[f"{tag}_res"]
    pad, lambdas = int(g["pad"][0]), g["lambdas"]
    rows, stride = ref.shape
    d_ref = torch.from_numpy(ref).cuda()
    planes = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_interp_planes_dev(d_ref.data_ptr(), stride, rows, stride, planes.data_ptr(), None))
    d_cur = torch.from_numpy(cur).cuda()
    nreq = len(req)
    reqs = (capi.MeRequest * nreq)()
    for i, r in enumerate(req):
        rq = reqs[i]
        rq.x0, rq.y0, rq.size = int(r[0]), int(r[1]), int(r[2])
        rq.mvp[0][0], rq.mvp[0][1], rq.mvp[1][0], rq.mvp[1][1] = int(r[3]), int(r[4]), int(r[5]), int(r[6])
        rq.zero_mv_tested = int(r[7])
        rq.mv_min_ipel[0], rq.mv_min_ipel[1], rq.mv_max_ipel[0], rq.mv_max_ipel[1] = int(r[8]), int(r[9]), int(r[10]), int(r[11])
        rq.sqrt_lambda = int(math.sqrt(float(lambdas[r[12]])))
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_res = torch.zeros(nreq * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_me_search_batch_dev(d_cur.data_ptr(), cur.strides[0], planes.data_ptr(), stride, rows, pad,
                                         d_req.data_ptr(), nreq, 32, d_res.data_ptr(), None))
    torch.cuda.synchronize()
    res = (capi.MeResult * nreq).from_buffer_copy(d_res.cpu().numpy().tobytes())
    for i in range(nreq):
        got = (res[i].mv[0], res[i].mv[1], res[i].best_cost, res[i].best_sad, res[i].mvp_idx)
        assert got == tuple(int(v) for v in want[i]), f"clip {tag} request {i}: gpu {got} reference {tuple(want[i])}"


@pytest.mark.parametrize("use_surface", [False, True])
def test_me_search_replays_encoder_trace(L, use_surface):
    """The kernel on the requests the reference's encoder issued while coding three P pictures (inputs and
    outputs recorded from a bit-identical encode, tests/golden/me_trace.npz): one batch per picture; computed
    integer candidates, and integer candidates read from the picture pair's SAD surface (range 64)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "me_trace.npz"))
    pad = int(g["pad"][0])
    for poc in g["pocs"]:
        cur, ref = np.ascontiguousarray(g[f"p{poc}_cur"]), np.ascontiguousarray(g[f"p{poc}_ref"])
        req, lam, want = g[f"p{poc}_req"], g[f"p{poc}_lambda"], g[f"p{poc}_res"]
        rows, stride = ref.shape
        d_ref = torch.from_numpy(ref).cuda()
        planes = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
        capi.check(L.xmi_interp_planes_dev(d_ref.data_ptr(), stride, rows, stride, planes.data_ptr(), None))
        d_cur = torch.from_numpy(cur).cuda()
        nreq = len(req)
        reqs = (capi.MeRequest * nreq)()
        for i, r in enumerate(req):
            rq = reqs[i]
            rq.x0, rq.y0, rq.size = int(r[0]), int(r[1]), int(r[2])
            rq.mvp[0][0], rq.mvp[0][1], rq.mvp[1][0], rq.mvp[1][1] = int(r[3]), int(r[4]), int(r[5]), int(r[6])
            rq.zero_mv_tested = int(r[7])
            rq.mv_min_ipel[0], rq.mv_min_ipel[1], rq.mv_max_ipel[0], rq.mv_max_ipel[1] = int(r[8]), int(r[9]), int(r[10]), int(r[11])
            rq.sqrt_lambda = int(math.sqrt(float(lam[i])))
        d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
        d_res = torch.zeros(nreq * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
        h, w = cur.shape
        surface = make_surface(L, d_cur, d_ref, w, h, stride, pad, 64) if use_surface else None
        search(L, d_cur, cur.strides[0], planes, stride, rows, pad, d_req, nreq, d_res, surface)
        res = (capi.MeResult * nreq).from_buffer_copy(d_res.cpu().numpy().tobytes())
        for i in range(nreq):
            got = (res[i].mv[0], res[i].mv[1], res[i].best_cost, res[i].best_sad, res[i].mvp_idx)
            assert got == tuple(int(v) for v in want[i]), f"picture {poc} request {i}: gpu {got} encoder {tuple(want[i])}"


def test_me_search_empty_and_errors(L):
    assert L.xmi_me_search_batch_dev(None, 0, None, 0, 0, 0, None, 0, 32, None, None) == 0
    a = torch.zeros(64, dtype=torch.uint8, device="cuda")
    assert L.xmi_me_search_batch_dev(a.data_ptr(), 6, a.data_ptr(), 8, 8, 0, a.data_ptr(), 1, 32, a.data_ptr(), None) == -1
    bad = capi.SadGeom(100, 64, 100, 200, 64)      # not a CTU multiple
    assert L.xmi_me_search_surface_batch_dev(a.data_ptr(), 8, a.data_ptr(), 8, 8, 0, C.byref(bad), a.data_ptr(), a.data_ptr(), a.data_ptr(),
                                             a.data_ptr(), a.data_ptr(), 1, 32, a.data_ptr(), None) == -1


def test_requests_that_would_leave_the_buffers_are_refused(L):
    """The request table is device memory the host side never sees, so the kernel checks each request: a PU outside the
    picture, a size that is no PU size, or predictors / a clip box reaching beyond the border are answered with
    mvp_idx -1 and cost 0xffffffff; the requests around them are served."""
    w, h = 128, 64
    cur, ref = make_video(w, h, 4, (1, 1), 2)
    rows, stride = ref.shape
    d_ref = torch.from_numpy(ref).cuda()
    planes = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_interp_planes_dev(d_ref.data_ptr(), stride, rows, stride, planes.data_ptr(), None))
    d_cur = torch.from_numpy(cur).cuda()
    reqs = (capi.MeRequest * 7)()
    for rq in reqs:
        rq.size, rq.x0, rq.y0, rq.sqrt_lambda = 16, 32, 16, 4
        rq.mv_min_ipel[0], rq.mv_min_ipel[1], rq.mv_max_ipel[0], rq.mv_max_ipel[1] = -64, -64, 64, 64
    reqs[1].size = 24                                       # no PU size
    reqs[2].x0 = w - 8                                      # sticks out of the picture
    reqs[3].y0 = -16
    reqs[4].mvp[0][0] = 4 * 300                             # predictor far outside the border
    reqs[5].mv_max_ipel[1] = 500                            # clip box far outside the border
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    for surface in (None, make_surface(L, d_cur, d_ref, w, h, stride, PAD, 64)):
        d_res = torch.zeros(7 * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
        search(L, d_cur, cur.strides[0], planes, stride, rows, PAD, d_req, 7, d_res, surface)
        res = (capi.MeResult * 7).from_buffer_copy(d_res.cpu().numpy().tobytes())
        for i in (1, 2, 3, 4, 5):
            assert (res[i].mvp_idx, res[i].best_cost, res[i].best_sad) == (-1, 0xffffffff, 0xffffffff), f"request {i} was served"
        assert res[0].mvp_idx == 0 and res[0].best_cost < 0x7fffffff
        assert bytes(res[0]) == bytes(res[6])
    # the host-table form says which request it refuses
    f = C.c_void_p()
    capi.check(L.xmi_frame_create(w, h, PAD, C.byref(f)))
    try:
        capi.check(L.xmi_frame_begin(f, O.p(cur), cur.strides[0], O.p(ref), stride))
        out = (capi.MeResult * 7)()
        assert L.xmi_frame_me_search(f, reqs, 1, 32, out) == 0
        assert bytes(out[0]) == bytes(res[0])
        one = (capi.MeRequest * 1)(reqs[4])
        assert L.xmi_frame_me_search(f, one, 1, 32, out) == -1 and b"request 0" in L.xmi_last_error()
    finally:
        L.xmi_frame_destroy(f)
