#!/usr/bin/env python3
"""Secondary kernels: TU chain and block metrics, device-resident batches, HIP-event timing.
Prints one JSON line per kernel/size with the HBM roofline fraction (algorithmic bytes: 5N^2+4 per TU,
2N^2+4 per metric pair; peak 8 TB/s)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x265_amd import capi  # noqa: E402

L = capi.load()
capi.check(L.xmi_init(0))
PEAK = 8000.0


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


ONLY = os.environ.get("XMI_BENCH_ONLY", "")     # "", "tu", "metric" or "me"
g = torch.Generator(device="cuda").manual_seed(1)
for n in (4, 8, 16, 32) if ONLY in ("", "tu") else ():
    cnt = (256 << 20) // (n * n)
    pix = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device="cuda", generator=g)
    pred = (pix.to(torch.int16) + torch.randint(-12, 13, pix.shape, dtype=torch.int16, device="cuda", generator=g)).clamp(0, 255).to(torch.uint8)
    qp = torch.full((cnt,), 27, dtype=torch.uint8, device="cuda")
    fl = torch.zeros(cnt, dtype=torch.uint8, device="cuda")
    lev = torch.empty((cnt, n, n), dtype=torch.int16, device="cuda")
    rec = torch.empty((cnt, n, n), dtype=torch.uint8, device="cuda")
    sm = torch.empty(cnt, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: capi.check(L.xmi_tu_chain_batch_dev(n, cnt, pix.data_ptr(), pred.data_ptr(), qp.data_ptr(), fl.data_ptr(),
                                                           lev.data_ptr(), rec.data_ptr(), sm.data_ptr(), s)))
    by = cnt * (5 * n * n + 4)
    print(json.dumps({"kernel": f"tu_chain {n}x{n}", "count": cnt, "ms": round(t * 1e3, 3), "MTU_per_s": round(cnt / t / 1e6, 1),
                      "GBps": round(by / t / 1e9, 1), "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))
    # the four stages on their own (xSubDct, xQuant, xDeQuant, xIDctAdd batches)
    coef = torch.empty((cnt, n, n), dtype=torch.int16, device="cuda")
    stages = (
        ("sub_dct", 4 * n * n, lambda: L.xmi_sub_dct_batch_dev(n, cnt, pix.data_ptr(), pred.data_ptr(), fl.data_ptr(), coef.data_ptr(), s)),
        ("quant", 4 * n * n + 4, lambda: L.xmi_quant_batch_dev(n, cnt, coef.data_ptr(), qp.data_ptr(), fl.data_ptr(), lev.data_ptr(), sm.data_ptr(), s)),
        ("dequant", 4 * n * n, lambda: L.xmi_dequant_batch_dev(n, cnt, lev.data_ptr(), qp.data_ptr(), coef.data_ptr(), s)),
        ("idct_add", 4 * n * n, lambda: L.xmi_idct_add_batch_dev(n, cnt, coef.data_ptr(), pred.data_ptr(), fl.data_ptr(), rec.data_ptr(), s)),
    )
    for name, bpt, fn in stages:
        t = timeit(lambda: capi.check(fn()))
        by = cnt * bpt
        print(json.dumps({"kernel": f"{name} {n}x{n}", "count": cnt, "ms": round(t * 1e3, 3), "MTU_per_s": round(cnt / t / 1e6, 1),
                          "GBps": round(by / t / 1e9, 1), "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))
    del pix, pred, lev, rec, coef
for metric, name in ((capi.METRIC_SAD, "sad"), (capi.METRIC_SATD_HM, "satd_hm"), (capi.METRIC_SATD_X264, "satd_x264"), (capi.METRIC_SSD, "ssd")):
    for n in (4, 8, 16, 32, 64) if ONLY in ("", "metric") else ():
        if metric == capi.METRIC_SATD_X264 and n > 16:
            continue
        cnt = (256 << 20) // (n * n)
        a = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device="cuda", generator=g)
        b = torch.randint(0, 256, (cnt, n, n), dtype=torch.uint8, device="cuda", generator=g)
        out = torch.empty(cnt, dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        t = timeit(lambda: capi.check(L.xmi_block_metric_batch_dev(metric, n, cnt, a.data_ptr(), b.data_ptr(), out.data_ptr(), s)))
        by = cnt * (2 * n * n + 4)
        print(json.dumps({"kernel": f"{name} {n}x{n}", "count": cnt, "ms": round(t * 1e3, 3), "Meval_per_s": round(cnt / t / 1e6, 1),
                          "GBps": round(by / t / 1e9, 1), "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))
        del a, b

if ONLY not in ("", "me"):
    sys.exit(0)
# ---- interpolation planes + batched motion-search replay on a 1080p frame pair ----
import ctypes as C  # noqa: E402
import math  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

W, H, PAD = 1920, 1088, 80
f0, f1 = B.synth_frame(W, H, 0, 7), B.synth_frame(W, H, 1, 7)
ref = torch.from_numpy(B.pad_frame(f0, PAD)).cuda()
cur = torch.from_numpy(f1).cuda()
rows, stride = ref.shape
planes = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
t = timeit(lambda: capi.check(L.xmi_interp_planes_dev(ref.data_ptr(), stride, rows, stride, planes.data_ptr(), s)))
by = rows * stride * 17
print(json.dumps({"kernel": "interp_planes 1080p (+pad 80)", "ms": round(t * 1e3, 3), "GBps": round(by / t / 1e9, 1),
                  "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))
# the same with rows 128 bytes apart (a wavefront's store is then two whole lines per plane), and on a 2160p frame
for tag, (w2, h2) in (("1080p", (1920, 1088)), ("2160p", (3840, 2176))):
    rows2, cols2 = h2 + 2 * PAD, w2 + 2 * PAD
    stride2 = (cols2 + 127) // 128 * 128
    ref2 = torch.randint(0, 256, (rows2, stride2), dtype=torch.uint8, device="cuda")
    planes2 = torch.zeros((16, rows2, stride2), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: capi.check(L.xmi_interp_planes_dev(ref2.data_ptr(), stride2, rows2, cols2, planes2.data_ptr(), s)))
    by2 = rows2 * cols2 * 17
    print(json.dumps({"kernel": f"interp_planes {tag} (+pad 80), stride {stride2}", "ms": round(t * 1e3, 3), "GBps": round(by2 / t / 1e9, 1),
                      "frac_hbm_peak": round(by2 / t / 1e9 / PEAK, 4)}))
    del ref2, planes2
rng = np.random.default_rng(3)
for size, nreq in ((8, 28000), (16, 9500), (32, 2400), (64, 600), (0, 39000)):
    reqs = (capi.MeRequest * nreq)()
    for i in range(nreq):
        n = size or int(rng.choice([8, 8, 8, 8, 8, 8, 16, 16, 32, 64]))
        rq = reqs[i]
        rq.size = n
        rq.x0 = int(rng.integers(0, (W - n) // 8 + 1)) * 8
        rq.y0 = int(rng.integers(0, (H - n) // 8 + 1)) * 8
        rq.mvp[0][0], rq.mvp[0][1] = int(rng.integers(-24, 25)), int(rng.integers(-24, 25))
        rq.mvp[1][0], rq.mvp[1][1] = int(rng.integers(-24, 25)), int(rng.integers(-24, 25))
        rq.zero_mv_tested = 0
        rq.mv_min_ipel[0], rq.mv_min_ipel[1], rq.mv_max_ipel[0], rq.mv_max_ipel[1] = -64, -64, 64, 64
        rq.sqrt_lambda = int(math.sqrt(0.578 * 2 ** ((22 - 12) / 3.0)))
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_res = torch.zeros(nreq * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: capi.check(L.xmi_me_search_batch_dev(cur.data_ptr(), W, planes.data_ptr(), stride, rows, PAD,
                                                            d_req.data_ptr(), nreq, 32, d_res.data_ptr(), s)), iters=5)
    res = (capi.MeResult * nreq).from_buffer_copy(d_res.cpu().numpy().tobytes())
    sads = sum(r.n_int_sads + r.n_sub_sads for r in res)
    print(json.dumps({"kernel": f"me_search {'mixed' if not size else str(size) + 'x' + str(size)}", "requests": nreq, "ms": round(t * 1e3, 3),
                      "Msearch_per_s": round(nreq / t / 1e6, 2), "sads_per_search": round(sads / nreq, 1),
                      "Msad_per_s": round(sads / t / 1e6, 1)}))

# ---- the same searches on quad-tree positions: integer candidates computed vs looked up in the resident SAD surface ----
from x265_amd import layout  # noqa: E402

R = 64
nctu = (W // 64) * (H // 64)
geom = capi.SadGeom(W, H, W, stride, R)
sp = [torch.zeros(layout.plane_elems(nctu, R, lv), dtype=torch.int16 if lv < 2 else torch.int32, device="cuda") for lv in range(4)]
capi.check(L.xmi_sad_surface_dev(C.byref(geom), cur.data_ptr(), ref.data_ptr() + PAD * stride + PAD, sp[0].data_ptr(), sp[1].data_ptr(),
                                 sp[2].data_ptr(), sp[3].data_ptr(), 0, nctu, s))
nreq = 39000
reqs = (capi.MeRequest * nreq)()
for i in range(nreq):
    n = int(rng.choice([8, 8, 8, 8, 8, 8, 16, 16, 32, 64]))
    rq = reqs[i]
    rq.size = n
    rq.x0 = int(rng.integers(0, W // n)) * n
    rq.y0 = int(rng.integers(0, H // n)) * n
    rq.mvp[0][0], rq.mvp[0][1] = int(rng.integers(-24, 25)), int(rng.integers(-24, 25))
    rq.mvp[1][0], rq.mvp[1][1] = int(rng.integers(-24, 25)), int(rng.integers(-24, 25))
    rq.zero_mv_tested = 0
    rq.mv_min_ipel[0], rq.mv_min_ipel[1], rq.mv_max_ipel[0], rq.mv_max_ipel[1] = -64, -64, 64, 64
    rq.sqrt_lambda = int(math.sqrt(0.578 * 2 ** ((22 - 12) / 3.0)))
d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
d_res = torch.zeros(nreq * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
d_res2 = torch.zeros_like(d_res)
t0 = timeit(lambda: capi.check(L.xmi_me_search_batch_dev(cur.data_ptr(), W, planes.data_ptr(), stride, rows, PAD,
                                                         d_req.data_ptr(), nreq, 32, d_res.data_ptr(), s)), iters=5)
t1 = timeit(lambda: capi.check(L.xmi_me_search_surface_batch_dev(cur.data_ptr(), W, planes.data_ptr(), stride, rows, PAD, C.byref(geom),
                                                                 sp[0].data_ptr(), sp[1].data_ptr(), sp[2].data_ptr(), sp[3].data_ptr(),
                                                                 d_req.data_ptr(), nreq, 32, d_res2.data_ptr(), s)), iters=5)
assert torch.equal(d_res, d_res2)
print(json.dumps({"kernel": "me_search mixed, quad-tree positions", "requests": nreq, "ms_computed": round(t0 * 1e3, 3),
                  "ms_surface_lookup": round(t1 * 1e3, 3)}))
del sp

# ---- intra mode search: every PU of a 1080p frame at each depth (35 modes, prediction + SATD) ----
frame = torch.from_numpy(f1).cuda()
for n in (4, 8, 16, 32):
    nx, ny = W // n, H // n
    nreq = nx * ny
    reqs = (capi.IntraRequest * nreq)()
    for i in range(nreq):
        rq = reqs[i]
        rq.x0, rq.y0, rq.size, rq.valids, rq.strong_smoothing = (i % nx) * n, (i // nx) * n, n, 0x1f, 1
        rq.mpm[0], rq.mpm[1], rq.mpm[2] = 0, 1, 26
        rq.sqrt_lambda = 4.7
    refs = torch.randint(0, 256, (nreq, capi.INTRA_REF_PITCH), dtype=torch.uint8, device="cuda", generator=g)
    d_req = torch.frombuffer(bytearray(bytes(reqs)), dtype=torch.uint8).cuda()
    d_res = torch.zeros(nreq * C.sizeof(capi.IntraResult), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: capi.check(L.xmi_intra_search_batch_dev(frame.data_ptr(), W, refs.data_ptr(), d_req.data_ptr(), nreq,
                                                               d_res.data_ptr(), s)), iters=5)
    print(json.dumps({"kernel": f"intra_search {n}x{n}", "requests": nreq, "ms": round(t * 1e3, 3),
                      "Msearch_per_s": round(nreq / t / 1e6, 2), "Gpixel_modes_per_s": round(nreq * n * n * 35 / t / 1e9, 1)}))

# ---- motion-data glue of a 1080p P picture: candidates (xFillMvpCand), request table, vector write-back for 32 640 8x8 PUs ----
npu8 = (W // 8) * (H // 8)
pu8 = (capi.MvpRequest * npu8)()
for i in range(npu8):
    pu8[i].x0, pu8[i].y0, pu8[i].size_x, pu8[i].size_y = (i % (W // 8)) * 8, (i // (W // 8)) * 8, 8, 8
d_pu8 = torch.frombuffer(bytearray(bytes(pu8)), dtype=torch.uint8).cuda()
d_cd8 = torch.zeros(npu8 * C.sizeof(capi.MvpResult), dtype=torch.uint8, device="cuda")
d_rq8 = torch.zeros(npu8 * C.sizeof(capi.MeRequest), dtype=torch.uint8, device="cuda")
d_rs8 = torch.zeros(npu8 * C.sizeof(capi.MeResult), dtype=torch.uint8, device="cuda")
vmap = torch.ones((H // 4, W // 4), dtype=torch.uint8, device="cuda")
mmap = torch.randint(-40, 41, (H // 8, W // 8, 2), device="cuda", generator=g).to(torch.int16)
for name, fn in (("mvp_candidates", lambda: L.xmi_mvp_candidates_dev(W, H, vmap.data_ptr(), mmap.data_ptr(), d_pu8.data_ptr(), npu8, d_cd8.data_ptr(), s)),
                 ("me_requests_from_candidates", lambda: L.xmi_me_requests_from_candidates_dev(W, H, d_pu8.data_ptr(), d_cd8.data_ptr(), None, npu8, 9, d_rq8.data_ptr(), s)),
                 ("mv_field_store", lambda: L.xmi_mv_field_store_dev(W, H, d_pu8.data_ptr(), d_rs8.data_ptr(), npu8, vmap.data_ptr(), mmap.data_ptr(), s))):
    t = timeit(lambda: capi.check(fn()))
    print(json.dumps({"kernel": f"{name} 8x8 PUs", "requests": npu8, "ms": round(t * 1e3, 4), "MPU_per_s": round(npu8 / t / 1e6, 1)}))

# ---- intra chroma mode search: every 8x8 chroma CU (16x16 luma) of a 1080p I picture ----
cw, chh = W // 2, H // 2
pu_ = torch.randint(0, 256, (chh, cw), dtype=torch.uint8, device="cuda", generator=g)
pv_ = torch.randint(0, 256, (chh, cw), dtype=torch.uint8, device="cuda", generator=g)
ncu = (cw // 8) * (chh // 8)
creq = (capi.ChromaRequest * ncu)()
for i in range(ncu):
    creq[i].x0, creq[i].y0, creq[i].size, creq[i].luma_mode = (i % (cw // 8)) * 8, (i // (cw // 8)) * 8, 8, int(rng.integers(0, 35))
d_creq = torch.frombuffer(bytearray(bytes(creq)), dtype=torch.uint8).cuda()
d_cref = torch.randint(0, 256, (ncu, 2, capi.INTRA_REF_PITCH), dtype=torch.uint8, device="cuda", generator=g)
d_cres = torch.zeros(ncu * C.sizeof(capi.ChromaResult), dtype=torch.uint8, device="cuda")
t = timeit(lambda: capi.check(L.xmi_intra_chroma_search_batch_dev(pu_.data_ptr(), pv_.data_ptr(), cw, d_cref.data_ptr(), d_creq.data_ptr(), ncu,
                                                                  d_cres.data_ptr(), s)))
print(json.dumps({"kernel": "intra_chroma_search 8x8", "requests": ncu, "ms": round(t * 1e3, 4), "Msearch_per_s": round(ncu / t / 1e6, 2)}))

# ---- in-loop deblocking of a 1080p picture: strengths as an I picture has them (2 on the 8x8 transform grid) and as a
# P picture has them (1 on a third of the segments) ----
for label, dens, val in (("I picture, strength 2 on every edge segment", 1.0, 2), ("P picture, strength 1 on a third of the segments", 0.33, 1)):
    yp = torch.from_numpy(f1).cuda().clone()
    up = torch.randint(100, 140, (H // 2, W // 2), dtype=torch.uint8, device="cuda", generator=g)
    vp = torch.randint(100, 140, (H // 2, W // 2), dtype=torch.uint8, device="cuda", generator=g)
    tab = (W // 4) * (H // 4)
    vb = ((torch.rand(tab, device="cuda", generator=g) < dens) * val).to(torch.uint8)
    hb = ((torch.rand(tab, device="cuda", generator=g) < dens) * val).to(torch.uint8)
    qpt = torch.full((tab,), 32, dtype=torch.uint8, device="cuda")
    prm = capi.DeblockParams(W, H, W, W // 2, 0, 0, 0, 0)
    t = timeit(lambda: capi.check(L.xmi_deblock_frame_dev(yp.data_ptr(), up.data_ptr(), vp.data_ptr(), C.byref(prm), vb.data_ptr(), hb.data_ptr(),
                                                          qpt.data_ptr(), s)))
    by = 2 * (W * H * 3 // 2) + 3 * tab
    print(json.dumps({"kernel": f"deblock 1080p ({label})", "ms": round(t * 1e3, 4), "GBps": round(by / t / 1e9, 1),
                      "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))

# ---- boundary strengths of a 1080p P picture from picture-level maps (16x16 units, a third with coefficients, a motion
# field with steps): reads unit size and cbf per 4x4 and a vector per 8x8, writes both strength tables ----
lgm = torch.full((H // 4, W // 4), 4, dtype=torch.uint8, device="cuda")
cbm = (torch.rand((H // 16, W // 16), device="cuda", generator=g) < 0.33).to(torch.uint8).repeat_interleave(4, 0).repeat_interleave(4, 1).contiguous()
mvm = torch.randint(-8, 9, (H // 16, W // 16, 2), device="cuda", generator=g).to(torch.int16).repeat_interleave(2, 0).repeat_interleave(2, 1).contiguous()
nbs = (W // 8) * (H // 4)
vb2, hb2 = torch.empty(nbs, dtype=torch.uint8, device="cuda"), torch.empty(nbs, dtype=torch.uint8, device="cuda")
t = timeit(lambda: capi.check(L.xmi_deblock_strengths_dev(W, H, 0, lgm.data_ptr(), cbm.data_ptr(), mvm.data_ptr(), vb2.data_ptr(), hb2.data_ptr(), s)))
by = 2 * (W // 4) * (H // 4) + 4 * (W // 8) * (H // 8) + 2 * nbs
print(json.dumps({"kernel": "deblock_strengths 1080p (P picture)", "ms": round(t * 1e3, 4), "GBps": round(by / t / 1e9, 1),
                  "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))

# ---- SAO statistics of a 1080p picture (every CTU and component, five types) ----
recs = [torch.from_numpy(f0).cuda(), torch.randint(0, 256, (H // 2, W // 2), dtype=torch.uint8, device="cuda", generator=g),
        torch.randint(0, 256, (H // 2, W // 2), dtype=torch.uint8, device="cuda", generator=g)]
orgs = [torch.from_numpy(f1).cuda(), recs[1].clone(), recs[2].clone()]
sg = capi.SaoGeom(W, H, W, W // 2, W, W // 2)
st_out = torch.zeros(nctu * 3 * C.sizeof(capi.SaoStats), dtype=torch.uint8, device="cuda")
t = timeit(lambda: capi.check(L.xmi_sao_stats_dev(C.byref(sg), recs[0].data_ptr(), recs[1].data_ptr(), recs[2].data_ptr(), orgs[0].data_ptr(),
                                                  orgs[1].data_ptr(), orgs[2].data_ptr(), st_out.data_ptr(), s)))
by = 2 * (W * H * 3 // 2) + st_out.numel()
print(json.dumps({"kernel": "sao_stats 1080p", "ms": round(t * 1e3, 4), "GBps": round(by / t / 1e9, 1), "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))
# ---- SAO offsets applied to the three planes of a 1080p picture (every CTU with an edge class or a band offset) ----
prm_tab = np.zeros((nctu, 8), np.uint8)
prm_tab[:, 0] = rng.integers(0, 5, nctu)
prm_tab[:, 1] = rng.integers(0, 32, nctu)
prm_tab[:, 2:6] = rng.integers(-3, 4, (nctu, 4)).astype(np.int8).view(np.uint8)
d_prm = torch.from_numpy(prm_tab).cuda()
dsts = [torch.zeros_like(r) for r in recs]


def sao_apply():
    for c in range(3):
        capi.check(L.xmi_sao_apply_plane_dev(recs[c].data_ptr(), W >> (c > 0), dsts[c].data_ptr(), W >> (c > 0), W >> (c > 0), H >> (c > 0),
                                             64 >> (c > 0), d_prm.data_ptr(), s))


t = timeit(sao_apply)
by = 2 * (W * H * 3 // 2)
print(json.dumps({"kernel": "sao_apply 1080p (Y, U, V)", "ms": round(t * 1e3, 4), "GBps": round(by / t / 1e9, 1), "frac_hbm_peak": round(by / t / 1e9 / PEAK, 4)}))
