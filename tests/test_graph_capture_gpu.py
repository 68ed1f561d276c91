"""GPU: the device entry points allocate nothing and synchronise nothing, so a picture's launches can be captured into
one HIP graph and replayed on new frame contents: SAD surface + TU chain + block metrics, replay equals eager."""
import ctypes as C

import numpy as np
import pytest

from x265_amd import capi, layout

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_capture_and_replay_matches_eager():
    L = capi.load()
    capi.check(L.xmi_init(0))
    W, H, R, PAD = 256, 128, 16, 32
    nctu = (W // 64) * (H // 64)
    g = torch.Generator(device="cuda").manual_seed(5)
    cur = torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda", generator=g)
    ref = torch.randint(0, 256, (H + 2 * PAD, W + 2 * PAD), dtype=torch.uint8, device="cuda", generator=g)
    stride = W + 2 * PAD
    geom = capi.SadGeom(W, H, W, stride, R)
    sp = [torch.zeros(layout.plane_elems(nctu, R, lv), dtype=torch.int16 if lv < 2 else torch.int32, device="cuda") for lv in range(4)]
    n, cnt = 8, 4096
    pix = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device="cuda", generator=g)
    pred = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device="cuda", generator=g)
    qp = torch.full((cnt,), 30, dtype=torch.uint8, device="cuda")
    fl = torch.zeros(cnt, dtype=torch.uint8, device="cuda")
    lev = torch.zeros((cnt, n, n), dtype=torch.int16, device="cuda")
    rec = torch.zeros((cnt, n, n), dtype=torch.uint8, device="cuda")
    sm = torch.zeros(cnt, dtype=torch.int32, device="cuda")
    sad = torch.zeros(cnt, dtype=torch.int32, device="cuda")

    def launch(stream):
        capi.check(L.xmi_sad_surface_dev(C.byref(geom), cur.data_ptr(), ref.data_ptr() + PAD * stride + PAD, sp[0].data_ptr(),
                                         sp[1].data_ptr(), sp[2].data_ptr(), sp[3].data_ptr(), 0, nctu, stream))
        capi.check(L.xmi_tu_chain_batch_dev(n, cnt, pix.data_ptr(), pred.data_ptr(), qp.data_ptr(), fl.data_ptr(), lev.data_ptr(),
                                            rec.data_ptr(), sm.data_ptr(), stream))
        capi.check(L.xmi_block_metric_batch_dev(capi.METRIC_SATD_HM, n, cnt, pix.data_ptr(), rec.data_ptr(), sad.data_ptr(), stream))

    outs = sp + [lev, rec, sm, sad]
    launch(torch.cuda.current_stream().cuda_stream)            # eager once (also sets one-time function attributes)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launch(torch.cuda.current_stream().cuda_stream)
    for trial in range(2):
        cur.copy_(torch.randint(0, 256, cur.shape, dtype=torch.uint8, device="cuda", generator=g))
        pix.copy_(torch.randint(0, 256, pix.shape, dtype=torch.uint8, device="cuda", generator=g))
        for o in outs:
            o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        replayed = [o.clone() for o in outs]
        for o in outs:
            o.zero_()
        launch(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for a, b in zip(replayed, outs):
            assert torch.equal(a, b), f"trial {trial}: graph replay differs from eager"
        assert int(sad.sum()) > 0
