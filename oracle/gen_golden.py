#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own compiled code (TEST INFRASTRUCTURE).

Runs only where /root/reference exists: `make -C oracle ref` compiles the reference's
src/pixel.cpp + src/table.cpp into oracle/_ref/libx265ref_pixel.so, and this script calls the
reference functions through ctypes on seeded inputs and stores {inputs, expected outputs}.
The fixtures are data only; no reference source text is stored.

Functions called (mangled C++ names; signatures from inc/x265.h:650-670 and src/pixel.cpp):
  xSadN[0..4]                    src/pixel.cpp:711
  xCalcHADs4x4/8x8/16x16/32x32/64x64, xCalcHADs32x32_test   src/pixel.cpp:63-527
  xSubDct / xQuant / xDeQuant / xIDctAdd                      src/pixel.cpp:3446, 1054, 1244, 5938
  xPredInterLumaBlk                                           src/pixel.cpp:8745
  xPredInterChromaBlk                                         src/pixel.cpp:9657
  xMotionSearch (from oracle/_ref/libx265ref_me.so, `make -C oracle ref_me`)   src/encode.cpp:2815
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
LIB = os.path.join(HERE, "_ref", "libx265ref_pixel.so")
LIB_ME = os.path.join(HERE, "_ref", "libx265ref_me.so")

SLICE_P, SLICE_I, MODE_INVALID = 1, 2, 255


def aligned(shape, dtype, align=64):
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    raw = np.zeros(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def ptr(a):
    return C.c_void_p(a.ctypes.data)


class XMV(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16)]


def edge_blocks(n):
    """(src, ref) pairs that hit the extremes: all-equal, max difference, checkerboards, DC-max."""
    z = np.zeros((n, n), np.uint8)
    f = np.full((n, n), 255, np.uint8)
    yy, xx = np.mgrid[0:n, 0:n]
    chk = (((yy + xx) & 1) * 255).astype(np.uint8)
    ramp = ((xx * 255) // max(n - 1, 1)).astype(np.uint8)
    return [(z, z), (f, f), (f, z), (z, f), (chk, 255 - chk), (chk, z), (ramp, ramp.T.copy()), (f, chk)]


def main():
    if not os.path.exists(LIB):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    lib = C.CDLL(LIB)
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20251111)

    def block_cases(n, n_random):
        cases = edge_blocks(n)
        for i in range(n_random):
            a = rng.integers(0, 256, (n, n), dtype=np.uint8)
            if i % 3 == 0:      # small residual, like a good prediction
                b = np.clip(a.astype(int) + rng.integers(-6, 7, (n, n)), 0, 255).astype(np.uint8)
            elif i % 3 == 1:    # smooth content
                b = np.clip(128 + 60 * np.sin(np.arange(n)[None, :] / 3.0 + i) + rng.integers(-9, 10, (n, n)), 0, 255).astype(np.uint8)
            else:
                b = rng.integers(0, 256, (n, n), dtype=np.uint8)
            cases.append((a, b))
        return cases

    # ---------------- SAD (xSadN) and HM SATD ----------------
    sad_tab = (C.c_void_p * 5).in_dll(lib, "xSadN")
    sad_proto = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32)
    had = {4: "_Z12xCalcHADs4x4PKhjS0_j", 8: "_Z12xCalcHADs8x8PKhjS0_j", 16: "_Z14xCalcHADs16x16PKhjS0_j",
           32: "_Z14xCalcHADs32x32PKhjS0_j", 64: "_Z14xCalcHADs64x64PKhjS0_j"}
    out = {}
    for idx, n in enumerate((4, 8, 16, 32, 64)):
        cases = block_cases(n, 24)
        src = aligned((64, 64), np.uint8)
        ref = aligned((80, 96), np.uint8)
        fsad = sad_proto(sad_tab[idx])
        fhad = getattr(lib, had[n])
        fhad.restype = C.c_uint32
        fhad.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        A = np.stack([c[0] for c in cases])
        B = np.stack([c[1] for c in cases])
        sads, satds = [], []
        for a, b in cases:
            src[:n, :n] = a
            ref[:n, :n] = b
            sads.append(fsad(ptr(src), 64, ptr(ref), 96))
            satds.append(fhad(ptr(src), 64, ptr(ref), 96))
        out[f"a{n}"], out[f"b{n}"] = A, B
        out[f"sad{n}"] = np.array(sads, np.uint32)
        out[f"satd_hm{n}"] = np.array(satds, np.uint32)
        if n == 32:  # the entry actually bound as xSatdN[3]
            f32 = lib._Z19xCalcHADs32x32_testPhlS_l
            f32.restype = C.c_int
            f32.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
            t = []
            for a, b in cases:
                src[:n, :n] = a
                ref[:n, :n] = b
                t.append(f32(ptr(src), 64, ptr(ref), 96))
            out["satd_tab3_32"] = np.array(t, np.uint32)
    np.savez_compressed(os.path.join(OUT, "sad_satd.npz"), **out)

    # ---------------- transform chain ----------------
    fsub = lib._Z7xSubDctPsPhjS0_jS_S_ij
    fsub.restype = None
    fsub.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
    fq = lib._Z6xQuantPsS_jjii10xSliceType
    fq.restype = C.c_uint32
    fq.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]
    fdq = lib._Z8xDeQuantPsS_jjii
    fdq.restype = None
    fdq.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int]
    fidct = lib._Z8xIDctAddPhjPsjS_S0_S0_ij
    fidct.restype = None
    fidct.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]

    S = 64
    pix = aligned((S, S), np.uint8)
    pred = aligned((S, S), np.uint8)
    rec = aligned((S, 80), np.uint8)
    t0 = aligned(((64 + 16) * 64,), np.int16)
    t1 = aligned(((64 + 16) * 64,), np.int16)
    coef = aligned((S, S), np.int16)
    deq = aligned((S, S), np.int16)
    out = {}
    qps = (0, 22, 27, 32, 37, 51)
    for n, mode, tag in ((4, MODE_INVALID, "dct4"), (4, 1, "dst4"), (8, MODE_INVALID, "dct8"),
                         (16, MODE_INVALID, "dct16"), (32, MODE_INVALID, "dct32")):
        cases = block_cases(n, 10)
        A = np.stack([c[0] for c in cases])
        B = np.stack([c[1] for c in cases])
        tr = np.zeros((len(cases), n, n), np.int16)
        lv = np.zeros((len(cases), len(qps), 2, n, n), np.int16)
        sm = np.zeros((len(cases), len(qps), 2), np.uint32)
        dq = np.zeros((len(cases), len(qps), 2, n, n), np.int16)
        rc = np.zeros((len(cases), len(qps), 2, n, n), np.uint8)
        for ci, (a, b) in enumerate(cases):
            pix[:n, :n] = a
            pred[:n, :n] = b
            t0[:] = 0
            fsub(ptr(t0), ptr(pix), S, ptr(pred), S, ptr(t0), ptr(t1), n, mode)
            tcoef = t0[:S * S].reshape(S, S)[:n, :n].copy()
            tr[ci] = tcoef
            for qi, qp in enumerate(qps):
                for si, st in enumerate((SLICE_P, SLICE_I)):
                    t0[:] = 0
                    t0[:S * S].reshape(S, S)[:n, :n] = tcoef
                    coef[:] = 0
                    sm[ci, qi, si] = fq(ptr(coef), ptr(t0), S, qp, n, n, st)
                    lv[ci, qi, si] = coef[:n, :n]
                    deq[:] = 0
                    fdq(ptr(deq), ptr(coef), S, qp, n, n)
                    dq[ci, qi, si] = deq[:n, :n]
                    rec[:] = 0
                    # same argument pattern as src/encode.cpp:3012 (tmp0 = psTmp[1], tmp1 = psTmp[0] = src)
                    t0[:] = 0
                    t0[:S * S].reshape(S, S)[:n, :n] = deq[:n, :n]
                    fidct(ptr(rec), 80, ptr(t0), S, ptr(pred), ptr(t1), ptr(t0), n, mode)
                    rc[ci, qi, si] = rec[:n, :n]
        out[f"{tag}_pix"], out[f"{tag}_pred"] = A, B
        out[f"{tag}_coef"], out[f"{tag}_level"], out[f"{tag}_sum"] = tr, lv, sm
        out[f"{tag}_deq"], out[f"{tag}_rec"] = dq, rc
    out["qps"] = np.array(qps, np.int32)
    np.savez_compressed(os.path.join(OUT, "transform.npz"), **out)

    # ---------------- quant / dequant / idct on raw coefficient inputs (incl. saturation) ----------------
    out = {}
    for n in (4, 8, 16, 32):
        cs = []
        for i in range(6):
            amp = (40, 400, 4000, 20000, 32640, 32767)[i]
            cs.append(rng.integers(-amp, amp + 1, (n, n)).astype(np.int16))
        cs.append(np.full((n, n), 32767, np.int16))
        cs.append(np.full((n, n), -32767, np.int16))
        Cs = np.stack(cs)
        lv = np.zeros((len(cs), len(qps), 2, n, n), np.int16)
        sm = np.zeros((len(cs), len(qps), 2), np.uint32)
        dq = np.zeros((len(cs), len(qps), n, n), np.int16)
        for ci, c in enumerate(cs):
            for qi, qp in enumerate(qps):
                for si, st in enumerate((SLICE_P, SLICE_I)):
                    t0[:] = 0
                    t0[:S * S].reshape(S, S)[:n, :n] = c
                    coef[:] = 0
                    sm[ci, qi, si] = fq(ptr(coef), ptr(t0), S, qp, n, n, st)
                    lv[ci, qi, si] = coef[:n, :n]
                # dequant of the raw values treated as levels (exercises the int16 saturation)
                coef[:] = 0
                coef[:n, :n] = c
                deq[:] = 0
                fdq(ptr(deq), ptr(coef), S, qp, n, n)
                dq[ci, qi] = deq[:n, :n]
        out[f"q{n}_in"], out[f"q{n}_level"], out[f"q{n}_sum"], out[f"q{n}_deq"] = Cs, lv, sm, dq
        # inverse transform of raw coefficient blocks (exercises the inter-stage clipping)
        for mode, tag in (((MODE_INVALID, "idct"),) if n != 4 else ((MODE_INVALID, "idct"), (1, "idst"))):
            pr = rng.integers(0, 256, (len(cs), n, n), dtype=np.uint8)
            rc = np.zeros((len(cs), n, n), np.uint8)
            for ci, c in enumerate(cs):
                pred[:] = 0
                pred[:n, :n] = pr[ci]
                rec[:] = 0
                t0[:] = 0
                t0[:S * S].reshape(S, S)[:n, :n] = c
                fidct(ptr(rec), 80, ptr(t0), S, ptr(pred), ptr(t1), ptr(t0), n, mode)
                rc[ci] = rec[:n, :n]
            out[f"{tag}{n}_pred"], out[f"{tag}{n}_rec"] = pr, rc
    out["qps"] = np.array(qps, np.int32)
    np.savez_compressed(os.path.join(OUT, "quant_raw.npz"), **out)

    # ---------------- luma interpolation ----------------
    fint = lib._Z17xPredInterLumaBlkPKhPhjjj3xMVPs
    fint.restype = None
    fint.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, XMV, C.c_void_p]
    out = {}
    RS = 112
    refbuf = aligned((96, RS), np.uint8)
    dst = aligned((64, 64), np.uint8)
    tmp = aligned(((64 + 16) * 64,), np.int16)
    for n in (8, 16, 32, 64):
        frames = rng.integers(0, 256, (2, 96, RS), dtype=np.uint8)
        frames[1] = np.where(rng.random((96, RS)) < 0.5, 0, 255).astype(np.uint8)  # extreme ringing
        res = np.zeros((2, 16, n, n), np.uint8)
        mvs = np.zeros((16, 2), np.int16)
        for fi in range(2):
            refbuf[:] = frames[fi]
            for fr in range(16):
                mv = XMV(int(4 * (fr % 3 - 1) + (fr & 3)), int(4 * ((fr // 5) - 1) + (fr >> 2)))
                mvs[fr] = (mv.x, mv.y)
                dst[:] = 0
                origin = C.c_void_p(refbuf.ctypes.data + 16 * RS + 16)
                fint(origin, ptr(dst), RS, n, n, mv, ptr(tmp))
                res[fi, fr] = dst[:n, :n]
        out[f"ref{n}"], out[f"mv{n}"], out[f"pred{n}"] = frames, mvs, res
    out["origin"] = np.array([16, 16], np.int32)
    np.savez_compressed(os.path.join(OUT, "interp.npz"), **out)
    if os.path.exists(LIB_ME):
        gen_me()
        gen_intra()
    else:
        print("skipping me_search.npz: make -C oracle ref_me first")
    print("golden vectors written to", os.path.normpath(OUT))


def gen_me():
    """Motion-search known answers: the reference's compiled xMotionSearch (src/encode.cpp:2815-3003, with the
    reference's ME_RANGE, MV_CLIP, MVD_COST_BY_TABLE and SATD_FOR_FRAC_ME settings) on seeded moving video.
    Stored per clip: cur, padded ref, request table and the four outputs."""
    lib = C.CDLL(LIB_ME)
    fme = lib._Z13xMotionSearchPKhS0_PsjjjPjP3xMVPiiS3_S3_dS2_
    fme.restype = XMV
    fme.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint, C.POINTER(C.c_uint32),
                    C.POINTER(XMV), C.POINTER(C.c_int), C.c_int, XMV, XMV, C.c_double, C.POINTER(C.c_uint)]
    PAD = 80
    out = {}
    lambdas = [0.578 * 2 ** ((qp - 12) / 3.0) for qp in (22, 27, 32, 37)] + [400.0, 2500.0]
    clips = (("a", 192, 128, 11, (3, 2), 6, 200), ("b", 128, 128, 12, (-7, 5), 2, 120), ("c", 128, 64, 13, (0, 0), 0, 60))
    for tag, w, h, seed, motion, noise, nreq in clips:
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:h + 2 * PAD, 0:w + 2 * PAD]
        base = 128 + 60 * np.sin(xx / 17.0) + 40 * np.cos(yy / 11.0) + 25 * np.sin((xx + 2 * yy) / 5.0)
        ref = aligned(base.shape, np.uint8)
        ref[:] = np.clip(base + rng.integers(-noise, noise + 1, base.shape), 0, 255).astype(np.uint8)
        dx, dy = motion
        cur = np.clip(ref[PAD + dy:PAD + dy + h, PAD + dx:PAD + dx + w].astype(int)
                      + rng.integers(-noise, noise + 1, (h, w)), 0, 255).astype(np.uint8)
        # request columns: x0 y0 size mvp0x mvp0y mvp1x mvp1y zero_tested minx miny maxx maxy lambda_index
        req = np.zeros((nreq, 13), np.int32)
        res = np.zeros((nreq, 5), np.int64)   # mvx mvy best_cost best_sad mvp_idx
        blk = aligned((64, 64), np.uint8)
        tmp = aligned(((64 + 16) * 64,), np.int16)
        for i in range(nreq):
            n = int(rng.choice([8, 16, 32, 64]))
            x0 = int(rng.integers(0, (w - n) // 8 + 1)) * 8
            y0 = int(rng.integers(0, (h - n) // 8 + 1)) * 8
            near = (-4 * dx + int(rng.integers(-6, 7)), -4 * dy + int(rng.integers(-6, 7)))
            style = i % 4
            if style == 0:
                m0, m1 = near, (near[0] + int(rng.integers(-3, 4)), near[1] + int(rng.integers(-3, 4)))
            elif style == 1:
                m0, m1 = (int(rng.integers(-200, 201)), int(rng.integers(-200, 201))), near
            elif style == 2:
                m0, m1 = (0, 0), (0, 0)
            else:
                m0 = (int(rng.integers(-256, 257)), int(rng.integers(-256, 257)))
                m1 = (int(rng.integers(-256, 257)), int(rng.integers(-256, 257)))
            zero = int(rng.integers(0, 2))
            lim = 64 if i % 7 else int(rng.integers(6, 40))
            mn = (-lim, -(64 if i % 5 else max(lim // 2, 5)))
            mx = (lim, 64 if i % 3 else lim)
            li = i % len(lambdas)
            req[i] = (x0, y0, n, m0[0], m0[1], m1[0], m1[1], zero, mn[0], mn[1], mx[0], mx[1], li)
            blk[:] = 0
            blk[:n, :n] = cur[y0:y0 + n, x0:x0 + n]
            mvp = (XMV * 2)(XMV(*m0), XMV(*m1))
            cost, sad, idx = C.c_uint32(0), C.c_uint(0), C.c_int(-7)
            origin = C.c_void_p(ref.ctypes.data + (PAD + y0) * ref.strides[0] + PAD + x0)
            mv = fme(ptr(blk), origin, ptr(tmp), ref.strides[0], n, n, C.byref(cost), mvp, C.byref(idx), zero,
                     XMV(*mn), XMV(*mx), lambdas[li], C.byref(sad))
            res[i] = (mv.x, mv.y, cost.value, sad.value, idx.value)
        out[f"{tag}_cur"], out[f"{tag}_ref"], out[f"{tag}_req"], out[f"{tag}_res"] = cur, np.array(ref), req, res
    out["lambdas"] = np.array(lambdas, np.float64)
    out["pad"] = np.array([PAD], np.int32)
    np.savez_compressed(os.path.join(OUT, "me_search.npz"), **out)



def gen_intra():
    """Intra luma prediction known answers from the reference's compiled code: xPaddingRef and xFilterRef
    (src/encode.cpp:3843, 3873), xEncIntraPred (src/pixel.cpp:7338, the SSE4 paths the encoder uses) for all 35 modes
    and sizes 4..32, and, for 32x32 (where the bound SATD is the reference's own C++ xCalcHADs32x32_test), the
    35-mode decision loop of src/encode.cpp:4021-4081 replayed in Python over those reference functions."""
    lib = C.CDLL(LIB_ME)
    fpad = lib._Z11xPaddingRefPhhPKi
    fpad.restype = None
    fpad.argtypes = [C.c_void_p, C.c_uint8, C.POINTER(C.c_int32)]
    ffil = lib._Z10xFilterRefPhS_ij
    ffil.restype = None
    ffil.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint]
    fprd = lib._Z13xEncIntraPredPhS_jjjj
    fprd.restype = None
    fprd.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    fsatd32 = lib._Z19xCalcHADs32x32_testPhlS_l
    fsatd32.restype = C.c_int
    fsatd32.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    ftype = (C.c_uint8 * (5 * 35)).in_dll(lib, "xg_aucIntraFilterType")
    rng = np.random.default_rng(77)
    out = {}
    valid_sets = (0x1f, 0x1e, 0x0e, 0x0c, 0x06, 0x18, 0x02, 0x08, 0x1a, 0x0a, 0x04)
    for li, n in enumerate((4, 8, 16, 32)):
        ncase = {4: 14, 8: 12, 16: 10, 32: 8}[n]
        L = 4 * n + 1
        raw = np.zeros((ncase, L), np.uint8)
        valids = np.zeros(ncase, np.uint8)
        strong = np.zeros(ncase, np.uint8)
        padded = np.zeros((ncase, L), np.uint8)
        filt = np.zeros((ncase, L), np.uint8)
        preds = np.zeros((ncase, 35, n, n), np.uint8)
        src = np.zeros((ncase, n, n), np.uint8)
        mpm = np.zeros((ncase, 3), np.uint8)
        lam = np.zeros(ncase, np.float64)
        satd = np.zeros((ncase, 35), np.uint32)
        best = np.zeros((ncase, 2), np.int32)       # mode, pred_idx
        cost = np.zeros(ncase, np.float64)
        ref0 = aligned((4 * 64 + 1 + 64,), np.uint8)
        ref1 = aligned((4 * 64 + 1 + 64,), np.uint8)
        dst = aligned((32 * 32 + 64,), np.uint8)
        pix = aligned((64, 64), np.uint8)
        offs = (C.c_int32 * 6)(0, n, 2 * n, 2 * n + 1, 3 * n + 1, 4 * n + 1)
        for ci in range(ncase):
            kind = ci % 4
            if kind == 0:
                r = rng.integers(0, 256, L)
            elif kind == 1:      # smooth ramp: exercises the bilinear (strong) filter decision at 32x32
                r = np.clip(100 + np.arange(L) * (0.2 + 0.1 * ci) + rng.integers(-1, 2, L), 0, 255)
            elif kind == 2:      # extremes
                r = np.where(rng.random(L) < 0.5, 0, 255)
            else:
                r = np.clip(128 + 90 * np.sin(np.arange(L) / (3.0 + ci)) + rng.integers(-5, 6, L), 0, 255)
            raw[ci] = r.astype(np.uint8)
            valids[ci] = valid_sets[ci % len(valid_sets)]
            strong[ci] = (ci >> 1) & 1
            ref0[:L] = raw[ci]
            fpad(ptr(ref0), int(valids[ci]), offs)
            padded[ci] = ref0[:L]
            ffil(ptr(ref1), ptr(ref0), n, int(strong[ci]))
            filt[ci] = ref1[:L]
            s = np.clip(128 + 70 * np.sin(np.add.outer(np.arange(n) * (0.3 + 0.05 * ci), np.arange(n) * 0.2)) + rng.integers(-20, 21, (n, n)), 0, 255)
            src[ci] = s.astype(np.uint8)
            pix[:] = 0
            pix[:n, :n] = src[ci]
            mpm[ci] = ((0, 1, 26), (10, 26, 0), (5, 4, 6), (34, 2, 33), (1, 1, 1))[ci % 5]
            lam[ci] = np.sqrt(0.57 * 2 ** ((np.array((22, 27, 32, 37))[ci % 4] - 12) / 3.0))
            bestc, bestm = np.finfo(np.float64).max, 0
            for mode in range(35):
                rf = ref1 if ftype[li * 35 + mode] else ref0
                dst[:] = 0
                fprd(ptr(dst), ptr(rf), n, mode, n, 1 if n != 32 else 0)
                preds[ci, mode] = dst[:n * n].reshape(n, n)
                if n == 32:
                    sv = fsatd32(ptr(pix), 64, ptr(dst), n)
                    satd[ci, mode] = sv
                    c = (1.0 if mode == mpm[ci, 0] else 2.0 if mode in (mpm[ci, 1], mpm[ci, 2]) else 3.0) * lam[ci]
                    c += float(sv)
                    if c < bestc:
                        bestc, bestm = c, mode
            if n == 32:
                m = [int(v) for v in mpm[ci]]
                if bestm in m:
                    idx = m.index(bestm)
                else:
                    sm = sorted(m)
                    idx = bestm
                    idx -= idx > sm[2]
                    idx -= idx > sm[1]
                    idx -= idx > sm[0]
                    idx += 3
                best[ci] = (bestm, idx)
                cost[ci] = bestc
        t = f"n{n}_"
        out[t + "raw"], out[t + "valids"], out[t + "strong"], out[t + "padded"], out[t + "filtered"] = raw, valids, strong, padded, filt
        out[t + "pred"], out[t + "src"], out[t + "mpm"], out[t + "sqrt_lambda"] = preds, src, mpm, lam
        if n == 32:
            out[t + "satd"], out[t + "best"], out[t + "cost"] = satd, best, cost
    np.savez_compressed(os.path.join(OUT, "intra.npz"), **out)


def gen_chroma():
    """Chroma prediction known answers: the reference's compiled xPredInterChromaBlk (src/pixel.cpp:9657, SSE4 paths)
    for block sizes 4..32, all 64 eighth-pel phases, on random and on 0/255 checkerboard-noise frames."""
    lib = C.CDLL(LIB)

    class XMV(C.Structure):
        _fields_ = [("x", C.c_int16), ("y", C.c_int16)]

    f = lib._Z19xPredInterChromaBlkPKhPhjjj3xMVPs
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, XMV, C.c_void_p]
    rng = np.random.default_rng(20240611)
    RS = 80
    refbuf = aligned((64, RS), np.uint8)
    dst = aligned((32, 32), np.uint8)
    tmp = aligned(((64 + 16) * 64,), np.int16)
    out = {}
    for n in (4, 8, 16, 32):
        frames = rng.integers(0, 256, (2, 64, RS), dtype=np.uint8)
        frames[1] = np.where(rng.random((64, RS)) < 0.5, 0, 255).astype(np.uint8)
        res = np.zeros((2, 64, n, n), np.uint8)
        mvs = np.zeros((64, 2), np.int16)
        for fi in range(2):
            refbuf[:] = frames[fi]
            for ph in range(64):
                mv = XMV(int(8 * (ph % 3 - 1) + (ph & 7)), int(8 * ((ph // 7) % 3 - 1) + (ph >> 3)))
                mvs[ph] = (mv.x, mv.y)
                dst[:] = 0
                origin = C.c_void_p(refbuf.ctypes.data + 12 * RS + 12)
                f(origin, ptr(dst), RS, n, n, mv, ptr(tmp))
                res[fi, ph] = dst[:n, :n]
        out[f"ref{n}"], out[f"mv{n}"], out[f"pred{n}"] = frames, mvs, res
    out["origin"] = np.array([12, 12], np.int32)
    np.savez_compressed(os.path.join(OUT, "interp_chroma.npz"), **out)
    print("wrote interp_chroma.npz")


def gen_intra_chroma():
    """xEncIntraPred (src/pixel.cpp:7338) called the way the chroma path calls it (bIsLuma = 0: no DC / horizontal /
    vertical edge filters) for sizes 4..32 and all 35 modes, next to the bIsLuma = 1 output for sizes 4..16."""
    lib = C.CDLL(LIB)
    fprd = lib._Z13xEncIntraPredPhS_jjjj
    fprd.restype = None
    fprd.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    rng = np.random.default_rng(4242)
    out = {}
    ref = aligned((4 * 64 + 1 + 64,), np.uint8)
    dst = aligned((32 * 32 + 64,), np.uint8)
    for n in (4, 8, 16, 32):
        L = 4 * n + 1
        ncase = 6
        refs = np.zeros((ncase, L), np.uint8)
        preds = np.zeros((ncase, 2, 35, n, n), np.uint8)
        for ci in range(ncase):
            if ci % 3 == 0:
                r = rng.integers(0, 256, L)
            elif ci % 3 == 1:
                r = np.where(rng.random(L) < 0.5, 0, 255)
            else:
                r = np.clip(128 + 100 * np.sin(np.arange(L) / (2.0 + ci)) + rng.integers(-4, 5, L), 0, 255)
            refs[ci] = r.astype(np.uint8)
            ref[:L] = refs[ci]
            for luma in (0, 1):
                if luma and n == 32:
                    continue          # the encoder never asks for it (src/encode.cpp:4038 passes nSize != 32)
                for mode in range(35):
                    dst[:] = 0
                    fprd(ptr(dst), ptr(ref), n, mode, n, luma)
                    preds[ci, luma, mode] = dst[:n * n].reshape(n, n)
        out[f"n{n}_ref"], out[f"n{n}_pred"] = refs, preds
    np.savez_compressed(os.path.join(OUT, "intra_pred_flags.npz"), **out)
    print("wrote intra_pred_flags.npz")


def gen_mvp():
    """MV predictor and merge candidate lists: the reference's compiled xFillMvpCand (src/encode.cpp:2703-2809) on seeded
    picture-level maps. For every request the CTU-cache neighbourhoods the function reads (mode flags with pitch MAX_PU_XY + 2,
    motion vectors with pitch MAX_MV_XY + 2) are cut out of a map of coded 4x4 blocks and of one motion vector per 8x8 block;
    stored: the maps, the requests, the two AMVP candidates, the merge candidates and their number."""
    lib = C.CDLL(LIB_ME)
    f = lib._Z12xFillMvpCandP3xMVPKhPKS_jjjj
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(41)
    W, H = 256, 128
    valid = (rng.random((H // 4, W // 4)) < 0.8).astype(np.uint8)
    valid[:, : W // 16] = 1                                   # a fully coded region
    valid[H // 8:, W // 8 * 3: W // 8 * 5] = 0                  # a region not coded yet
    palette = rng.integers(-40, 41, (5, 2)).astype(np.int16)  # few distinct vectors: the pruning comparisons fire
    palette[0] = 0
    mv = palette[rng.integers(0, 5, (H // 8, W // 8))]
    nreq = 600
    req = np.zeros((nreq, 6), np.int32)                       # x0 y0 size_x size_y pu_mode pu_idx
    res = np.zeros((nreq, 7, 2), np.int16)
    cnt = np.zeros(nreq, np.int32)
    for i in range(nreq):
        n = int(rng.choice([8, 16, 32, 64]))
        mode = int(rng.choice([0, 0, 0, 1, 2]))               # SIZE_2Nx2N, SIZE_2NxN, SIZE_Nx2N
        idx = int(rng.integers(0, 2)) if mode else 0
        sx, sy = (n, n) if mode == 0 or n == 8 else ((n, n // 2) if mode == 1 else (n // 2, n))
        if n == 8:
            mode = idx = 0
        x0 = int(rng.integers(0, (W - n) // 8 + 1)) * 8 + (sx if (mode == 2 and idx == 1) else 0)
        y0 = int(rng.integers(0, (H - n) // 8 + 1)) * 8 + (sy if (mode == 1 and idx == 1) else 0)
        req[i] = (x0, y0, sx, sy, mode, idx)
        modes = np.full((18, 18), 255, np.uint8)              # origin at [1][1]: rows / columns -1 .. 16 in 4x4 units
        for r in range(-1, 17):
            for c in range(-1, 17):
                yy, xx = y0 // 4 + r, x0 // 4 + c
                if 0 <= yy < H // 4 and 0 <= xx < W // 4 and valid[yy, xx]:
                    modes[r + 1, c + 1] = 0
        mvs = np.zeros((10, 10, 2), np.int16)                  # origin at [1][1]: -1 .. 8 in 8x8 units
        for r in range(-1, 9):
            for c in range(-1, 9):
                yy, xx = y0 // 8 + r, x0 // 8 + c
                if 0 <= yy < H // 8 and 0 <= xx < W // 8:
                    mvs[r + 1, c + 1] = mv[yy, xx]
        out = np.full((7, 2), 0x7777, np.int16)
        cnt[i] = f(ptr(out), C.c_void_p(modes.ctypes.data + 18 + 1), C.c_void_p(mvs.ctypes.data + (10 + 1) * 4), sx, sy, mode, idx)
        out[2 + cnt[i]:] = 0                                   # entries past the count are not written by the reference
        res[i] = out
    print("mvp: merge count histogram", np.bincount(cnt, minlength=6), "AMVP second candidate zero in", int((res[:, 1] == 0).all(axis=1).sum()))
    np.savez_compressed(os.path.join(OUT, "mvp_cand.npz"), valid=valid, mv=mv, req=req, cand=res, count=cnt, size=np.array([W, H], np.int32))
    print("wrote mvp_cand.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mvp":
        gen_mvp()
    elif len(sys.argv) > 1 and sys.argv[1] == "chroma":
        gen_chroma()
        gen_intra_chroma()
    else:
        main()
        gen_chroma()
        gen_intra_chroma()
        gen_mvp()
