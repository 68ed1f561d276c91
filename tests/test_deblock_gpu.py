"""GPU: in-loop deblocking (xmi_deblock_frame_dev / xmi_deblock_frame) against the pictures the reference's encoder
deblocked (tests/golden/deblock_trace.npz, recorded around xDeblockFrame in a bit-identical encode) and against the
oracle's CTB-ordered restatement on seeded pictures with every boundary strength, QP and offset."""
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


def gpu_deblock(L, y, u, v, prm, vbs, hbs, qp):
    dy, du, dv = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (y, u, v))
    dvb, dhb, dq = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (vbs, hbs, qp))
    capi.check(L.xmi_deblock_frame_dev(dy.data_ptr(), du.data_ptr(), dv.data_ptr(), C.byref(prm), dvb.data_ptr(), dhb.data_ptr(), dq.data_ptr(), None))
    torch.cuda.synchronize()
    return dy.cpu().numpy(), du.cpu().numpy(), dv.cpu().numpy()


def test_deblocking_replays_encoder_trace(L):
    g = np.load(os.path.join(GOLDEN, "deblock_trace.npz"))
    for poc in g["pocs"]:
        w, h, slice_, tco, bo, cb, cr = (int(x) for x in g[f"p{poc}_meta"])
        prm = capi.DeblockParams(w, h, w, w // 2, tco, bo, cb, cr)
        y, u, v = gpu_deblock(L, g[f"p{poc}_y0"], g[f"p{poc}_u0"], g[f"p{poc}_v0"], prm, g[f"p{poc}_vbs"], g[f"p{poc}_hbs"], g[f"p{poc}_qp"])
        for k, a in zip("yuv", (y, u, v)):
            np.testing.assert_array_equal(a, g[f"p{poc}_{k}1"], err_msg=f"picture {poc} plane {k}")
        # host-pointer form on planes with the encoder's kind of strides (padding around the picture stays untouched)
        pad = 24
        Y = np.full((h + 2 * pad, w + 2 * pad + 4), 99, np.uint8)
        U = np.full((h // 2 + pad, w // 2 + pad + 2), 98, np.uint8)
        V = U.copy()
        Y[pad:pad + h, pad:pad + w] = g[f"p{poc}_y0"]
        U[pad // 2:pad // 2 + h // 2, pad // 2:pad // 2 + w // 2] = g[f"p{poc}_u0"]
        V[pad // 2:pad // 2 + h // 2, pad // 2:pad // 2 + w // 2] = g[f"p{poc}_v0"]
        prm2 = capi.DeblockParams(w, h, Y.strides[0], U.strides[0], tco, bo, cb, cr)
        vbs, hbs, qp = (np.ascontiguousarray(g[f"p{poc}_{k}"]) for k in ("vbs", "hbs", "qp"))
        capi.check(L.xmi_deblock_frame(O.at(Y, pad, pad), O.at(U, pad // 2, pad // 2), O.at(V, pad // 2, pad // 2), C.byref(prm2), O.p(vbs), O.p(hbs), O.p(qp)))
        np.testing.assert_array_equal(Y[pad:pad + h, pad:pad + w], g[f"p{poc}_y1"])
        np.testing.assert_array_equal(U[pad // 2:pad // 2 + h // 2, pad // 2:pad // 2 + w // 2], g[f"p{poc}_u1"])
        np.testing.assert_array_equal(V[pad // 2:pad // 2 + h // 2, pad // 2:pad // 2 + w // 2], g[f"p{poc}_v1"])
        assert (Y[:pad] == 99).all() and (Y[:, :pad] == 99).all() and (Y[:, pad + w:] == 99).all() and (U[:pad // 2] == 98).all()


@pytest.mark.parametrize("w,h,seed", [(192, 128, 1), (448, 256, 2), (72, 40, 3), (1920, 1088, 4), (256, 64, 5), (64, 256, 6), (48, 48, 7)])
def test_deblocking_matches_oracle(L, w, h, seed):
    """Seeded pictures (smooth ramps with block steps so that weak, strong and no filtering all occur), random
    boundary strengths 0/1/2 on every segment, QP table 20..51, non-zero offsets. The sizes include pictures one CTB high
    and one CTB wide, of which xDeblockFrame filters the last CTB only (src/Deblock.cpp:512-559)."""
    orc = O.load()
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 50 * np.sin(xx / 37.0) + 40 * np.cos(yy / 23.0)
    steps = rng.integers(-12, 13, (h // 8 + 1, w // 8 + 1))
    y = np.clip(base + np.kron(steps, np.ones((8, 8)))[:h, :w] + rng.integers(-2, 3, (h, w)), 0, 255).astype(np.uint8)
    y[: h // 4] = rng.integers(0, 256, (h // 4, w), dtype=np.uint8)          # a noisy band: mostly "no filtering"
    cs = rng.integers(-10, 11, (h // 16 + 1, w // 16 + 1))
    u = np.clip(128 + np.kron(cs, np.ones((8, 8)))[: h // 2, : w // 2] + rng.integers(-1, 2, (h // 2, w // 2)), 0, 255).astype(np.uint8)
    v = np.clip(100 + 2 * np.kron(cs, np.ones((8, 8)))[: h // 2, : w // 2] + rng.integers(-3, 4, (h // 2, w // 2)), 0, 255).astype(np.uint8)
    tab = (w // 4) * (h // 4)
    vbs = rng.choice([0, 1, 2], tab, p=[0.3, 0.35, 0.35]).astype(np.uint8)
    hbs = rng.choice([0, 1, 2], tab, p=[0.3, 0.35, 0.35]).astype(np.uint8)
    qp = rng.integers(20, 52, (h // 4, w // 4)).astype(np.uint8)
    tco, bo, cb, cr = int(rng.integers(-6, 7)), int(rng.integers(-6, 7)), int(rng.integers(-4, 5)), int(rng.integers(-4, 5))
    wy, wu, wv = y.copy(), u.copy(), v.copy()
    oprm = O.OrcDeblockParams(w, h, w, w // 2, tco, bo, cb, cr)
    orc.orc_deblock_frame(O.p(wy), O.p(wu), O.p(wv), C.byref(oprm), O.p(vbs), O.p(hbs), O.p(qp))
    one_ctb = (w <= 64) != (h <= 64)      # only the last CTB of such a picture is filtered
    assert (wy != y).sum() > (40 if one_ctb else w * h // 50) and (wu != u).sum() > 0
    prm = capi.DeblockParams(w, h, w, w // 2, tco, bo, cb, cr)
    gy, gu, gv = gpu_deblock(L, y, u, v, prm, vbs, hbs, qp)
    np.testing.assert_array_equal(gy, wy)
    np.testing.assert_array_equal(gu, wu)
    np.testing.assert_array_equal(gv, wv)


def test_deblocking_errors(L):
    a = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    bad = capi.DeblockParams(20, 16, 20, 10, 0, 0, 0, 0)
    assert L.xmi_deblock_frame_dev(a.data_ptr(), a.data_ptr(), a.data_ptr(), C.byref(bad), a.data_ptr(), a.data_ptr(), a.data_ptr(), None) == -1
    ok = capi.DeblockParams(16, 16, 16, 8, 0, 0, 0, 0)
    assert L.xmi_deblock_frame_dev(None, a.data_ptr(), a.data_ptr(), C.byref(ok), a.data_ptr(), a.data_ptr(), a.data_ptr(), None) == -1


@pytest.mark.parametrize("w,h,pad", [(64, 48, 80), (448, 256, 80), (224, 128, 40), (33, 7, 5)])
def test_frame_padding_replicates_the_border(L, w, h, pad):
    """xmi_frame_padding_dev / xmi_frame_padding against numpy's edge replication (what xFramePadding,
    src/encode.cpp:2043-2092, writes: rows left and right, then whole rows up and down with the corners)."""
    rng = np.random.default_rng(w + h)
    pic = rng.integers(0, 256, (h, w), dtype=np.uint8)
    want = np.pad(pic, pad, mode="edge")
    stride = w + 2 * pad + 8
    buf = np.full((h + 2 * pad, stride), 123, np.uint8)
    buf[pad:pad + h, pad:pad + w] = pic
    d = torch.from_numpy(buf).cuda()
    capi.check(L.xmi_frame_padding_dev(d.data_ptr() + pad * stride + pad, stride, w, h, pad, None))
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    np.testing.assert_array_equal(got[:, :w + 2 * pad], want)
    assert (got[:, w + 2 * pad:] == 123).all()
    capi.check(L.xmi_frame_padding(O.at(buf, pad, pad), stride, w, h, pad))
    np.testing.assert_array_equal(buf[:, :w + 2 * pad], want)
    assert (buf[:, w + 2 * pad:] == 123).all()
    assert L.xmi_frame_padding_dev(d.data_ptr(), w, w, h, pad, None) == -1


def test_boundary_strengths_replay_encoder_trace(L):
    """xmi_deblock_strengths_dev / xmi_deblock_strengths from the picture-level maps recorded in bit-identical encodes
    (tests/golden/bs_trace.npz): the vertical_bs / horizontal_bs arrays the reference's per-unit calls had filled."""
    g = np.load(os.path.join(GOLDEN, "bs_trace.npz"))
    for t in g["tags"]:
        w, h, slice_ = (int(v) for v in g[f"{t}_meta"])
        intra = int(slice_ == 2)
        lg, cbf, mv = (np.ascontiguousarray(g[f"{t}_{k}"]) for k in ("unit_log2", "cbf", "mv"))
        n = (w // 8) * (h // 4)
        d_lg, d_cbf, d_mv = (torch.from_numpy(a).cuda() for a in (lg, cbf, mv))
        d_v = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        d_h = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        capi.check(L.xmi_deblock_strengths_dev(w, h, intra, d_lg.data_ptr(), d_cbf.data_ptr(), None if intra else d_mv.data_ptr(), d_v.data_ptr(),
                                               d_h.data_ptr(), None))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_v.cpu().numpy(), g[f"{t}_vbs"], err_msg=f"{t}: vertical edges")
        np.testing.assert_array_equal(d_h.cpu().numpy(), g[f"{t}_hbs"], err_msg=f"{t}: horizontal edges")
        vbs, hbs = np.full(n, 9, np.uint8), np.full(n, 9, np.uint8)
        capi.check(L.xmi_deblock_strengths(w, h, intra, O.p(lg), O.p(cbf), O.p(mv), O.p(vbs), O.p(hbs)))
        np.testing.assert_array_equal(vbs, g[f"{t}_vbs"])
        np.testing.assert_array_equal(hbs, g[f"{t}_hbs"])


@pytest.mark.parametrize("w,h,intra,seed", [(1920, 1088, 0, 1), (1920, 1088, 1, 2), (200, 72, 0, 3)])
def test_boundary_strengths_match_oracle(L, w, h, intra, seed):
    """Random quad-tree partitions (units of 4..64 aligned to their size, 32x32 quarters included), random cbf flags and a
    motion field with small and large steps between units, and inside units too (which the reference's decisions never
    produce but its code handles), against orc_deblock_strengths; picture sizes off the CTU grid included."""
    orc = O.load()
    rng = np.random.default_rng(seed)
    lg = np.zeros((h // 4, w // 4), np.uint8)
    mv = np.zeros((h // 8, w // 8, 2), np.int16)

    def split(x0, y0, size):
        if size > 4 and (x0 + size > w or y0 + size > h or (size > (4 if intra else 8) and rng.random() < 0.55)):
            for dy in (0, size // 2):
                for dx in (0, size // 2):
                    if x0 + dx < w and y0 + dy < h:
                        split(x0 + dx, y0 + dy, size // 2)
            return
        lg[y0 // 4:(y0 + size) // 4, x0 // 4:(x0 + size) // 4] = int(np.log2(size))
        if size >= 8:
            base = rng.integers(-40, 41, 2)
            mv[y0 // 8:(y0 + size) // 8, x0 // 8:(x0 + size) // 8] = base
            if size >= 16 and rng.random() < 0.3:         # a step inside the unit
                mv[y0 // 8 + 1:(y0 + size) // 8, x0 // 8 + 1:(x0 + size) // 8] += rng.integers(-6, 7, 2).astype(np.int16)
    for y0 in range(0, h, 64):
        for x0 in range(0, w, 64):
            split(x0, y0, 64)
    assert lg.min() >= 2
    cbf = (rng.random(lg.shape) < 0.3).astype(np.uint8)
    n = (w // 8) * (h // 4)
    want_v, want_h = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    orc.orc_deblock_strengths(w, h, intra, O.p(lg), O.p(cbf), O.p(mv), O.p(want_v), O.p(want_h))
    if not intra:
        assert (want_v == 1).sum() > n // 50 and (want_v == 0).sum() > n // 50
    d_lg, d_cbf, d_mv = (torch.from_numpy(a).cuda() for a in (lg, cbf, mv))
    d_v, d_h = torch.full((n,), 9, dtype=torch.uint8, device="cuda"), torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_deblock_strengths_dev(w, h, intra, d_lg.data_ptr(), d_cbf.data_ptr(), d_mv.data_ptr(), d_v.data_ptr(), d_h.data_ptr(), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_v.cpu().numpy(), want_v)
    np.testing.assert_array_equal(d_h.cpu().numpy(), want_h)
    assert L.xmi_deblock_strengths_dev(w + 4, h, intra, d_lg.data_ptr(), d_cbf.data_ptr(), d_mv.data_ptr(), d_v.data_ptr(), d_h.data_ptr(), None) == -1
    assert L.xmi_deblock_strengths_dev(w, h, 0, d_lg.data_ptr(), d_cbf.data_ptr(), None, d_v.data_ptr(), d_h.data_ptr(), None) == -1
