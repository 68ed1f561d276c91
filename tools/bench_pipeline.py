#!/usr/bin/env python3
"""GPU-side work of one 1080p P picture as ONE captured HIP graph: quarter-pel planes of the previous reconstruction,
(optionally) the SAD surface, a picture's worth of motion searches, the prediction blocks of every PU (luma + both
chroma planes), the transform chains of every TU, then the in-loop filters on a reconstructed picture (boundary strengths from decision maps, deblocking,
SAO statistics, SAO offsets) and the border padding, all device-resident, requests and decisions synthetic. It measures what a host
CTU loop would have to keep fed - the decisions themselves (MVPs, splits, modes) are the host's and are given here.
Prints one JSON line per variant."""
import ctypes as C
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from x265_amd import capi, layout  # noqa: E402

L = capi.load()
capi.check(L.xmi_init(0))
W, H, PAD, R = 1920, 1088, 80, 64
rng = np.random.default_rng(11)
f0, f1 = B.synth_frame(W, H, 0, 7), B.synth_frame(W, H, 1, 7)
ref = torch.from_numpy(B.pad_frame(f0, PAD)).cuda()
cur = torch.from_numpy(f1).cuda()
rows, stride = ref.shape
cpad = PAD // 2
ref_u = torch.randint(0, 256, (H // 2 + 2 * cpad, W // 2 + 2 * cpad), dtype=torch.uint8, device="cuda")
ref_v = torch.randint(0, 256, ref_u.shape, dtype=torch.uint8, device="cuda")
cstride = ref_u.shape[1]
planes = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
nctu = (W // 64) * (H // 64)
geom = capi.SadGeom(W, H, W, stride, R)
sp = [torch.zeros(layout.plane_elems(nctu, R, lv), dtype=torch.int16 if lv < 2 else torch.int32, device="cuda") for lv in range(4)]

# motion searches: the mix of one picture (every 8x8 of a quarter of the CTUs, 16x16 / 32x32 / 64x64 of all)
reqs = []
for n, frac in ((64, 1.0), (32, 1.0), (16, 1.0), (8, 0.85)):
    for y in range(0, H, n):
        for x in range(0, W, n):
            if rng.random() < frac:
                reqs.append((x, y, n))
nreq = len(reqs)
me = (capi.MeRequest * nreq)()
for i, (x, y, n) in enumerate(reqs):
    rq = me[i]
    rq.x0, rq.y0, rq.size = x, y, n
    rq.mvp[0][0], rq.mvp[0][1] = -12 + int(rng.integers(-4, 5)), -8 + int(rng.integers(-4, 5))
    rq.mvp[1][0], rq.mvp[1][1] = -12 + int(rng.integers(-4, 5)), -8 + int(rng.integers(-4, 5))
    rq.mv_min_ipel[0], rq.mv_min_ipel[1], rq.mv_max_ipel[0], rq.mv_max_ipel[1] = -64, -64, 64, 64
    rq.sqrt_lambda = int(math.sqrt(0.578 * 2 ** ((32 - 12) / 3.0)))
d_me = torch.frombuffer(bytearray(bytes(me)), dtype=torch.uint8).cuda()
d_me_res = torch.zeros(nreq * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
# the same searches derived on the device: a motion field near the true motion and a map of coded blocks stand for the
# neighbours' decisions; candidates (xFillMvpCand) and request table are then built by two small kernels per picture
mv_field = torch.from_numpy((np.array([-12, -8]) + rng.integers(-4, 5, (H // 8, W // 8, 2))).astype(np.int16)).cuda()
coded = torch.ones((H // 4, W // 4), dtype=torch.uint8, device="cuda")
pus = (capi.MvpRequest * nreq)()
for i, (x, y, n) in enumerate(reqs):
    pus[i].x0, pus[i].y0, pus[i].size_x, pus[i].size_y = x, y, n, n
d_pus = torch.frombuffer(bytearray(bytes(pus)), dtype=torch.uint8).cuda()
d_cand = torch.zeros(nreq * C.sizeof(capi.MvpResult), dtype=torch.uint8, device="cuda")
d_me_built = torch.zeros(nreq * C.sizeof(capi.MeRequest), dtype=torch.uint8, device="cuda")

# final partition for prediction + transform: 16x16 PUs over the whole picture, one 16x16 luma TU and two 8x8 chroma TUs each
pus = [(x, y) for y in range(0, H, 16) for x in range(0, W, 16)]
npu = len(pus)
pr = (capi.InterPredRequest * npu)()
for i, (x, y) in enumerate(pus):
    pr[i].x0, pr[i].y0, pr[i].size = x, y, 16
    pr[i].mv[0], pr[i].mv[1] = -12 + int(rng.integers(-3, 4)), -8 + int(rng.integers(-3, 4))
    pr[i].out_y, pr[i].out_c = i * 256, i * 64
d_pr = torch.frombuffer(bytearray(bytes(pr)), dtype=torch.uint8).cuda()
pred_y = torch.zeros((npu, 16, 16), dtype=torch.uint8, device="cuda")
pred_u = torch.zeros((npu, 8, 8), dtype=torch.uint8, device="cuda")
pred_v = torch.zeros((npu, 8, 8), dtype=torch.uint8, device="cuda")
# source blocks in the same packed order (a gather the host loop's staging would do once per picture)
src_y = cur.reshape(H // 16, 16, W // 16, 16).permute(0, 2, 1, 3).contiguous().reshape(npu, 16, 16)
src_c = torch.randint(0, 256, (2 * npu, 8, 8), dtype=torch.uint8, device="cuda")
pred_c = torch.zeros((2 * npu, 8, 8), dtype=torch.uint8, device="cuda")
qp_y = torch.full((npu,), 32, dtype=torch.uint8, device="cuda")
qp_c = torch.full((2 * npu,), 31, dtype=torch.uint8, device="cuda")
fl_y = torch.zeros(npu, dtype=torch.uint8, device="cuda")
fl_c = torch.zeros(2 * npu, dtype=torch.uint8, device="cuda")
lev_y = torch.zeros((npu, 16, 16), dtype=torch.int16, device="cuda")
rec_y = torch.zeros((npu, 16, 16), dtype=torch.uint8, device="cuda")
sum_y = torch.zeros(npu, dtype=torch.int32, device="cuda")
lev_c = torch.zeros((2 * npu, 8, 8), dtype=torch.int16, device="cuda")
rec_c = torch.zeros((2 * npu, 8, 8), dtype=torch.uint8, device="cuda")
sum_c = torch.zeros(2 * npu, dtype=torch.int32, device="cuda")

# in-loop filters and padding work on a padded reconstruction of their own (the scatter of the TU chains' blocks into
# it is the host loop's bookkeeping and is not modelled)
rec_pad = torch.from_numpy(B.pad_frame(f1, PAD)).cuda()
rec_u = torch.randint(100, 140, (H // 2 + 2 * cpad, W // 2 + 2 * cpad), dtype=torch.uint8, device="cuda")
rec_v = torch.randint(100, 140, rec_u.shape, dtype=torch.uint8, device="cuda")
sao_y, sao_u, sao_v = torch.zeros_like(rec_pad), torch.zeros_like(rec_u), torch.zeros_like(rec_v)
org_u = torch.randint(100, 140, (H // 2, W // 2), dtype=torch.uint8, device="cuda")
org_v = torch.randint(100, 140, (H // 2, W // 2), dtype=torch.uint8, device="cuda")
tab = (W // 4) * (H // 4)
vbs, hbs = torch.zeros(tab, dtype=torch.uint8, device="cuda"), torch.zeros(tab, dtype=torch.uint8, device="cuda")
# decision maps the boundary strengths are derived from: 16x16 units, a third with coefficients, a motion field with steps
unit_map = torch.full((H // 4, W // 4), 4, dtype=torch.uint8, device="cuda")
cbf_map = (torch.rand((H // 16, W // 16), device="cuda") < 0.33).to(torch.uint8).repeat_interleave(4, 0).repeat_interleave(4, 1).contiguous()
mv_map = torch.randint(-8, 9, (H // 16, W // 16, 2), device="cuda").to(torch.int16).repeat_interleave(2, 0).repeat_interleave(2, 1).contiguous()
qpt = torch.full((tab,), 32, dtype=torch.uint8, device="cuda")
dbp = capi.DeblockParams(W, H, stride, cstride, 0, 0, 0, 0)
sgeo = capi.SaoGeom(W, H, stride, cstride, W, W // 2)
sao_stats = torch.zeros(nctu * 3 * C.sizeof(capi.SaoStats), dtype=torch.uint8, device="cuda")
prm_tab = np.zeros((nctu, 8), np.uint8)
prm_tab[:, 0] = rng.integers(0, 5, nctu)
prm_tab[:, 2:6] = rng.integers(-2, 3, (nctu, 4)).astype(np.int8).view(np.uint8)
d_prm = torch.from_numpy(prm_tab).cuda()
y00, u00, v00 = PAD * stride + PAD, cpad * cstride + cpad, cpad * cstride + cpad


def filters(stream):
    capi.check(L.xmi_deblock_strengths_dev(W, H, 0, unit_map.data_ptr(), cbf_map.data_ptr(), mv_map.data_ptr(), vbs.data_ptr(), hbs.data_ptr(), stream))
    capi.check(L.xmi_deblock_frame_dev(rec_pad.data_ptr() + y00, rec_u.data_ptr() + u00, rec_v.data_ptr() + v00, C.byref(dbp), vbs.data_ptr(),
                                       hbs.data_ptr(), qpt.data_ptr(), stream))
    capi.check(L.xmi_sao_stats_dev(C.byref(sgeo), rec_pad.data_ptr() + y00, rec_u.data_ptr() + u00, rec_v.data_ptr() + v00, cur.data_ptr(),
                                   org_u.data_ptr(), org_v.data_ptr(), sao_stats.data_ptr(), stream))
    for src, dst, o, st, ww, hh, cs, pd in ((rec_pad, sao_y, y00, stride, W, H, 64, PAD), (rec_u, sao_u, u00, cstride, W // 2, H // 2, 32, cpad),
                                            (rec_v, sao_v, v00, cstride, W // 2, H // 2, 32, cpad)):
        capi.check(L.xmi_sao_apply_plane_dev(src.data_ptr() + o, st, dst.data_ptr() + o, st, ww, hh, cs, d_prm.data_ptr(), stream))
        capi.check(L.xmi_frame_padding_dev(dst.data_ptr() + o, st, ww, hh, pd, stream))


def picture(stream, surface):
    capi.check(L.xmi_interp_planes_dev(ref.data_ptr(), stride, rows, stride, planes.data_ptr(), stream))
    capi.check(L.xmi_mvp_candidates_dev(W, H, coded.data_ptr(), mv_field.data_ptr(), d_pus.data_ptr(), nreq, d_cand.data_ptr(), stream))
    capi.check(L.xmi_me_requests_from_candidates_dev(W, H, d_pus.data_ptr(), d_cand.data_ptr(), None, nreq, int(math.sqrt(0.578 * 2 ** ((32 - 12) / 3.0))),
                                                     d_me_built.data_ptr(), stream))
    if surface:
        capi.check(L.xmi_sad_surface_dev(C.byref(geom), cur.data_ptr(), ref.data_ptr() + PAD * stride + PAD, sp[0].data_ptr(), sp[1].data_ptr(),
                                         sp[2].data_ptr(), sp[3].data_ptr(), 0, nctu, stream))
        capi.check(L.xmi_me_search_surface_batch_dev(cur.data_ptr(), W, planes.data_ptr(), stride, rows, PAD, C.byref(geom), sp[0].data_ptr(),
                                                     sp[1].data_ptr(), sp[2].data_ptr(), sp[3].data_ptr(), d_me_built.data_ptr(), nreq, 32,
                                                     d_me_res.data_ptr(), stream))
    else:
        capi.check(L.xmi_me_search_batch_dev(cur.data_ptr(), W, planes.data_ptr(), stride, rows, PAD, d_me_built.data_ptr(), nreq, 32,
                                             d_me_res.data_ptr(), stream))
    capi.check(L.xmi_pred_inter_batch_dev(ref.data_ptr() + PAD * stride + PAD, stride, ref_u.data_ptr() + cpad * cstride + cpad,
                                          ref_v.data_ptr() + cpad * cstride + cpad, cstride, d_pr.data_ptr(), npu, pred_y.data_ptr(),
                                          pred_u.data_ptr(), pred_v.data_ptr(), stream))
    capi.check(L.xmi_tu_chain_batch_dev(16, npu, src_y.data_ptr(), pred_y.data_ptr(), qp_y.data_ptr(), fl_y.data_ptr(), lev_y.data_ptr(),
                                        rec_y.data_ptr(), sum_y.data_ptr(), stream))
    # the two chroma planes: 8x8 TUs
    capi.check(L.xmi_tu_chain_batch_dev(8, npu, src_c.data_ptr(), pred_u.data_ptr(), qp_c.data_ptr(), fl_c.data_ptr(), lev_c.data_ptr(),
                                        rec_c.data_ptr(), sum_c.data_ptr(), stream))
    capi.check(L.xmi_tu_chain_batch_dev(8, npu, src_c.data_ptr() + npu * 64, pred_v.data_ptr(), qp_c.data_ptr() + npu, fl_c.data_ptr() + npu,
                                        lev_c.data_ptr() + npu * 128, rec_c.data_ptr() + npu * 64, sum_c.data_ptr() + npu * 4, stream))
    filters(stream)


for surface in (False, True):
    s = torch.cuda.current_stream().cuda_stream
    picture(s, surface)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        picture(torch.cuda.current_stream().cuda_stream, surface)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 50
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    t_graph = a.elapsed_time(b) / iters
    a.record()
    for _ in range(iters):
        picture(s, surface)
    b.record()
    torch.cuda.synchronize()
    t_eager = a.elapsed_time(b) / iters
    print(json.dumps({"pipeline": "1080p P picture: planes + MV candidates and search requests of " + str(nreq) + " PUs" + (" + SAD surface (R=64)" if surface else "") + f" + {nreq} motion searches"
                      + (" (integer SADs from the surface)" if surface else "") + f" + {npu} PUs predicted (Y, U, V) + {3 * npu} TU chains + boundary strengths + deblocking + SAO statistics and offsets + padding",
                      "ms_graph": round(t_graph, 3), "ms_eager": round(t_eager, 3), "pictures_per_s_graph": round(1e3 / t_graph, 1),
                      "coded_levels_nonzero_tus": int((sum_y != 0).sum().item())}))
