"""GPU: prediction blocks through the C ABI - xPredInterLumaBlk / xPredInterChromaBlk / xEncIntraPred as batches on
device planes and as per-call host-pointer drop-ins - against the reference's stored outputs (tests/golden) and the
oracle on seeded frames."""
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_inter_batch_matches_oracle(L):
    """A frame's worth of mixed PUs: luma and both chroma blocks of every request, all phases, against
    orc_pred_inter_luma / orc_pred_inter_chroma reading the same padded planes."""
    orc = O.load()
    rng = np.random.default_rng(21)
    W, H, PAD = 192, 128, 80
    y = rng.integers(0, 256, (H + 2 * PAD, W + 2 * PAD), dtype=np.uint8)
    u = rng.integers(0, 256, (H // 2 + PAD, W // 2 + PAD), dtype=np.uint8)      # chroma planes padded by PAD/2
    v = np.where(rng.random(u.shape) < 0.5, 0, 255).astype(np.uint8)
    n = 300
    reqs = (capi.InterPredRequest * n)()
    oy = oc = 0
    for i in range(n):
        s = int(rng.choice([8, 16, 32, 64]))
        rq = reqs[i]
        rq.size = s
        rq.x0 = int(rng.integers(0, (W - s) // 8 + 1)) * 8
        rq.y0 = int(rng.integers(0, (H - s) // 8 + 1)) * 8
        rq.mv[0], rq.mv[1] = int(rng.integers(-272, 273)), int(rng.integers(-272, 273))   # up to 68 integer pels
        if i % 9 == 0:
            rq.mv[0] &= ~3
        if i % 11 == 0:
            rq.mv[1] &= ~7
        rq.out_y, rq.out_c = oy, oc
        oy += s * s
        oc += s * s // 4
    d_y, d_u, d_v = dev(y), dev(u), dev(v)
    py = torch.zeros(oy, dtype=torch.uint8, device="cuda")
    pu = torch.zeros(oc, dtype=torch.uint8, device="cuda")
    pv = torch.zeros(oc, dtype=torch.uint8, device="cuda")
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    sy, sc = y.strides[0], u.strides[0]
    capi.check(L.xmi_pred_inter_batch_dev(d_y.data_ptr() + PAD * sy + PAD, sy, d_u.data_ptr() + (PAD // 2) * sc + PAD // 2,
                                          d_v.data_ptr() + (PAD // 2) * sc + PAD // 2, sc, d_req.data_ptr(), n, py.data_ptr(),
                                          pu.data_ptr(), pv.data_ptr(), None))
    torch.cuda.synchronize()
    gy, gu, gv = py.cpu().numpy(), pu.cpu().numpy(), pv.cpu().numpy()
    by, bc = np.zeros((64, 64), np.uint8), np.zeros((32, 32), np.uint8)
    for i in range(n):
        rq = reqs[i]
        s, c = rq.size, rq.size // 2
        orc.orc_pred_inter_luma(O.at(y, PAD + rq.y0, PAD + rq.x0), O.p(by), sy, s, s, rq.mv[0], rq.mv[1])
        np.testing.assert_array_equal(gy[rq.out_y:rq.out_y + s * s].reshape(s, s), by[:s, :s], err_msg=f"luma request {i}")
        for plane, got in ((u, gu), (v, gv)):
            orc.orc_pred_inter_chroma(O.at(plane, PAD // 2 + rq.y0 // 2, PAD // 2 + rq.x0 // 2), O.p(bc), sc, c, c, rq.mv[0], rq.mv[1])
            np.testing.assert_array_equal(got[rq.out_c:rq.out_c + c * c].reshape(c, c), bc[:c, :c], err_msg=f"chroma request {i}")
    # luma only / chroma only
    py.zero_()
    capi.check(L.xmi_pred_inter_batch_dev(d_y.data_ptr() + PAD * sy + PAD, sy, None, None, 0, d_req.data_ptr(), n, py.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(py.cpu().numpy(), gy)
    assert L.xmi_pred_inter_batch_dev(None, 0, None, None, 0, d_req.data_ptr(), n, py.data_ptr(), None, None, None) == -1
    assert L.xmi_pred_inter_batch_dev(None, 0, None, None, 0, None, 0, None, None, None, None) == 0


@pytest.mark.parametrize("n", [8, 16, 32, 64])
def test_percall_luma_matches_reference_golden(L, n):
    g = np.load(os.path.join(GOLDEN, "interp.npz"))
    oy, ox = g["origin"]
    frames, mvs, want = g[f"ref{n}"], g[f"mv{n}"], g[f"pred{n}"]
    for fi in range(frames.shape[0]):
        ref = np.ascontiguousarray(frames[fi])
        for k in range(len(mvs)):
            dst = np.zeros((64, 64), np.uint8)
            capi.check(L.xmi_pred_inter_luma(O.at(ref, oy, ox), O.p(dst), ref.strides[0], n, n, int(mvs[k, 0]), int(mvs[k, 1])))
            np.testing.assert_array_equal(dst[:n, :n], want[fi, k], err_msg=f"frame {fi} mv {tuple(mvs[k])}")
            assert not dst[:, n:].any() and not dst[n:, :].any()


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_percall_chroma_matches_reference_golden(L, n):
    g = np.load(os.path.join(GOLDEN, "interp_chroma.npz"))
    oy, ox = g["origin"]
    frames, mvs, want = g[f"ref{n}"], g[f"mv{n}"], g[f"pred{n}"]
    for fi in range(frames.shape[0]):
        ref = np.ascontiguousarray(frames[fi])
        for k in range(len(mvs)):
            dst = np.zeros((32, 32), np.uint8)
            capi.check(L.xmi_pred_inter_chroma(O.at(ref, oy, ox), O.p(dst), ref.strides[0], n, n, int(mvs[k, 0]), int(mvs[k, 1])))
            np.testing.assert_array_equal(dst[:n, :n], want[fi, k], err_msg=f"frame {fi} mv {tuple(mvs[k])}")


def test_percall_inter_reads_only_what_the_filters_read(L):
    """An integer MV copies the block and nothing else: the block sits in the last rows/columns of its buffer."""
    rng = np.random.default_rng(3)
    ref = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    dst = np.zeros((64, 64), np.uint8)
    capi.check(L.xmi_pred_inter_luma(O.at(ref, 8, 8), O.p(dst), 16, 8, 8, 0, 0))
    np.testing.assert_array_equal(dst[:8, :8], ref[8:, 8:])
    capi.check(L.xmi_pred_inter_luma(O.at(ref, 4, 8), O.p(dst), 16, 8, 8, 0, 2))      # vertical only: rows 1..15, columns 8..15
    assert L.xmi_pred_inter_luma(O.p(ref), O.p(dst), 16, 8, 16, 0, 0) == -1
    assert L.xmi_pred_inter_chroma(O.p(ref), O.p(dst), 16, 64, 64, 0, 0) == -1


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_intra_pred_matches_reference_golden(L, n):
    """Batch and per-call xEncIntraPred against the reference's compiled function: every mode, with and without the
    luma edge filters."""
    g = np.load(os.path.join(GOLDEN, "intra_pred_flags.npz"))
    refs, want = g[f"n{n}_ref"], g[f"n{n}_pred"]
    cases = [(ci, luma, mode) for ci in range(len(refs)) for luma in (0, 1) if not (luma and n == 32) for mode in range(35)]
    reqs = (capi.IntraPredRequest * len(cases))()
    samples = np.zeros((len(cases), capi.INTRA_REF_PITCH), np.uint8)
    for i, (ci, luma, mode) in enumerate(cases):
        reqs[i].size, reqs[i].mode, reqs[i].is_luma, reqs[i].out = n, mode, luma, i * n * n
        samples[i, :4 * n + 1] = refs[ci]
    d_s = dev(samples)
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    out = torch.zeros(len(cases) * n * n, dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_intra_pred_batch_dev(d_s.data_ptr(), d_req.data_ptr(), len(cases), out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(len(cases), n, n)
    for i, (ci, luma, mode) in enumerate(cases):
        np.testing.assert_array_equal(got[i], want[ci, luma, mode], err_msg=f"case {ci} luma {luma} mode {mode}")
    # per-call form, destination stride 64 as the encoder's prediction buffers
    for i in range(0, len(cases), 7):
        ci, luma, mode = cases[i]
        dst = np.zeros((n, 64), np.uint8)
        rf = np.ascontiguousarray(refs[ci])
        capi.check(L.xmi_enc_intra_pred(O.p(dst), O.p(rf), 64, mode, n, luma))
        np.testing.assert_array_equal(dst[:, :n], want[ci, luma, mode])
        assert not dst[:, n:].any()
    assert L.xmi_enc_intra_pred(O.p(dst), O.p(rf), 64, 35, n, 1) == -1
    assert L.xmi_enc_intra_pred(O.p(dst), O.p(rf), 64, 0, 64, 1) == -1


def test_requests_with_impossible_sizes_write_nothing(L):
    """Request tables live in device memory, so the kernels guard themselves: a size that is no PU size is skipped."""
    y = torch.full((96, 96), 7, dtype=torch.uint8, device="cuda")
    reqs = (capi.InterPredRequest * 2)()
    reqs[0].size, reqs[0].x0, reqs[0].y0, reqs[0].out_y = 8, 16, 16, 0
    reqs[1].size, reqs[1].x0, reqs[1].y0, reqs[1].out_y = 1000, 16, 16, 64
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    out = torch.full((4096,), 255, dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_pred_inter_batch_dev(y.data_ptr(), 96, None, None, 0, d_req.data_ptr(), 2, out.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert (o[:64] == 7).all() and (o[64:] == 255).all()
    ir = (capi.IntraPredRequest * 1)()
    ir[0].size, ir[0].mode, ir[0].is_luma, ir[0].out = 64, 1, 1, 0
    d_ir = torch.frombuffer(bytearray(bytes(ir)), dtype=torch.uint8).cuda()
    samples = torch.zeros(capi.INTRA_REF_PITCH, dtype=torch.uint8, device="cuda")
    out.fill_(255)
    capi.check(L.xmi_intra_pred_batch_dev(samples.data_ptr(), d_ir.data_ptr(), 1, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert (out.cpu().numpy() == 255).all()
