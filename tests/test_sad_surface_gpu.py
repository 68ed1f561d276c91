"""GPU: frame-wide SAD surface (xmi_sad_surface*) against the CPU oracle, bit-exact, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O
from x265_amd import capi, layout

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = capi.load()
    capi.check(lib.xmi_init(0))
    return lib


@pytest.fixture(scope="module")
def orc():
    return O.load()


def make_frames(w, h, pad, seed, kind="noise"):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        cur = rng.integers(0, 256, (h, w), dtype=np.uint8)
        ref = rng.integers(0, 256, (h + 2 * pad, w + 2 * pad), dtype=np.uint8)
    elif kind == "extreme":   # maximum SADs: 64*64*255 needs the full u32 / u16 ranges
        cur = np.zeros((h, w), np.uint8)
        ref = np.full((h + 2 * pad, w + 2 * pad), 255, np.uint8)
    else:                      # smooth moving texture, like the benchmark's synthetic video
        yy, xx = np.mgrid[0:h + 2 * pad, 0:w + 2 * pad]
        base = 128 + 60 * np.sin((xx + 3) / 17.0) + 40 * np.cos((yy + 2) / 11.0)
        ref = np.clip(base + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8)
        cur = np.clip(base[pad + 2:pad + 2 + h, pad + 3:pad + 3 + w] + rng.integers(-6, 7, (h, w)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(cur), np.ascontiguousarray(ref)


def gpu_surface(L, cur, ref, pad, rng):
    """run the host-pointer entry point and return the four levels in natural [ctu][blk][dy][dx] order"""
    h, w = cur.shape
    g = capi.SadGeom(w, h, cur.strides[0], ref.strides[0], rng)
    nctu = (h // 64) * (w // 64)
    dts = (np.uint16, np.uint16, np.uint32, np.uint32)
    planes = []
    for lvl in range(4):
        n = L.xmi_sad_surface_elems(C.byref(g), lvl)
        assert n == layout.plane_elems(nctu, rng, lvl)
        planes.append(np.full(n, 0xAAAA, dts[lvl]))
    capi.check(L.xmi_sad_surface(C.byref(g), O.p(cur), O.at(ref, pad, pad), pad, *[O.p(x) for x in planes]))
    nat = [layout.untile(planes[lvl], nctu, rng, lvl) for lvl in range(4)]
    # the C-side index helper agrees with the host-side untiling
    r = np.random.default_rng(1)
    for lvl in range(4):
        for _ in range(20):
            ctu, blk = int(r.integers(nctu)), int(r.integers(layout.NBLK[lvl]))
            dy, dx = int(r.integers(-rng, rng + 1)), int(r.integers(-rng, rng + 1))
            idx = L.xmi_sad_surface_index(C.byref(g), lvl, ctu, blk, dy, dx)
            assert planes[lvl][idx] == nat[lvl][ctu, blk, dy + rng, dx + rng]
        idx = L.xmi_sad_surface_index(C.byref(g), lvl, nctu - 1, layout.NBLK[lvl] - 1, rng, rng)
        assert planes[lvl][idx] == nat[lvl][nctu - 1, -1, 2 * rng, 2 * rng]
    return nat


@pytest.mark.parametrize("w,h,rng,kind", [(64, 64, 16, "noise"), (128, 64, 32, "noise"), (64, 128, 64, "smooth"),
                                          (128, 128, 16, "extreme"), (64, 64, 64, "extreme"), (64, 64, 128, "noise")])
def test_surface_matches_oracle(L, orc, w, h, rng, kind):
    pad = ((rng + 4 + 15) // 16) * 16
    cur, ref = make_frames(w, h, pad, seed=w * 7 + h + rng, kind=kind)
    got = gpu_surface(L, cur, ref, pad, rng)
    want = O.sad_surface(orc, cur, ref, pad, rng)
    for name, a, b in zip(("sad8", "sad16", "sad32", "sad64"), got, want):
        np.testing.assert_array_equal(a, b, err_msg=f"{name} {w}x{h} R={rng} {kind}")


def test_surface_pyramid_property_416x240(L):
    """config 0 geometry (416x240 coded as 448x256, R=64): levels must sum exactly (no oracle needed)."""
    rng, pad = 64, 80
    cur, ref = make_frames(448, 256, pad, seed=1234, kind="smooth")
    s8, s16, s32, s64 = gpu_surface(L, cur, ref, pad, rng)
    nctu = s8.shape[0]
    a8 = s8.reshape(nctu, 8, 8, *s8.shape[2:]).astype(np.uint32)
    a16 = a8.reshape(nctu, 4, 2, 4, 2, *s8.shape[2:]).sum(axis=(2, 4))
    np.testing.assert_array_equal(a16.reshape(s16.shape), s16)
    a32 = a16.reshape(nctu, 2, 2, 2, 2, *s8.shape[2:]).sum(axis=(2, 4))
    np.testing.assert_array_equal(a32.reshape(s32.shape), s32)
    np.testing.assert_array_equal(a32.sum(axis=(1, 2)).reshape(s64.shape), s64)
    # zero-MV, direct numpy SAD of a few CTUs
    for ctu in (0, 6, 27):
        cy, cx = (ctu // 7) * 64, (ctu % 7) * 64
        d = np.abs(cur[cy:cy + 64, cx:cx + 64].astype(int) - ref[pad + cy:pad + cy + 64, pad + cx:pad + cx + 64].astype(int)).sum()
        assert s64[ctu, 0, rng, rng] == d
        d = np.abs(cur[cy:cy + 64, cx:cx + 64].astype(int) - ref[pad + cy - 64:pad + cy, pad + cx + 64:pad + cx + 128].astype(int)).sum()
        assert s64[ctu, 0, 0, 2 * rng] == d


def test_surface_rejects_bad_arguments(L):
    rng, pad = 16, 32
    cur, ref = make_frames(192, 64, pad, seed=3)
    h, w = cur.shape
    s = np.zeros(16, np.uint8)
    # bad arguments are rejected with a message, not executed
    bad = capi.SadGeom(w, h, cur.strides[0], ref.strides[0], 20)
    assert L.xmi_sad_surface(C.byref(bad), O.p(cur), O.at(ref, pad, pad), pad, O.p(s), O.p(s), O.p(s), O.p(s)) == -1
    assert b"range" in L.xmi_last_error()
    bad = capi.SadGeom(w - 2, h, cur.strides[0], ref.strides[0], rng)
    assert L.xmi_sad_surface(C.byref(bad), O.p(cur), O.at(ref, pad, pad), pad, O.p(s), O.p(s), O.p(s), O.p(s)) == -1
    assert b"multiples of 64" in L.xmi_last_error()
    good = capi.SadGeom(w, h, cur.strides[0], ref.strides[0], rng)
    assert L.xmi_sad_surface(C.byref(good), O.p(cur), O.at(ref, pad, pad), 8, O.p(s), O.p(s), O.p(s), O.p(s)) == -1
    assert b"pad" in L.xmi_last_error()


def _pyramid_checks(torch, planes, nctu, rng, ctu_lo, ctu_hi):
    """levels must sum exactly, checked on the device in the tiled layout: slot c*16+z of sad8 -> slot z of sad16,
    z = 4q+b -> quadrant q of sad32 -> sad64; main part [ctu][dy][tile][slot][16] and the dx = +R part [ctu][dy][slot]"""
    nmv, t = 2 * rng + 1, rng // 8
    pitch = [layout.row_pitch(rng, lvl) for lvl in range(4)]      # payload t * nb * 16 elements + padding nobody writes
    main = [nctu * nmv * pitch[lvl] for lvl in range(4)]
    for lo in range(ctu_lo, ctu_hi, 32):          # chunks of CTUs bound the int32 temporaries
        hi = min(lo + 32, ctu_hi)
        n = hi - lo

        def part(lvl, edge):
            nb = layout.NBLK[lvl]
            if edge:
                return planes[lvl][main[lvl] + lo * nmv * nb:main[lvl] + hi * nmv * nb].view(n, nmv, 1, nb, 1).to(torch.int32)
            rows = planes[lvl][lo * nmv * pitch[lvl]:hi * nmv * pitch[lvl]].view(n, nmv, pitch[lvl])
            if lvl == 0 and lo == ctu_lo:
                assert bool((rows[:, :, t * nb * 16:].to(torch.int32) == 0xAAAA).all()), "the padding between rows was written"
            return rows[:, :, :t * nb * 16].reshape(n, nmv, t, nb, 16).to(torch.int32)

        for edge in (False, True):
            s8, s16, s32, s64 = (part(lvl, edge) for lvl in range(4))
            w = s8.shape[-1]
            tt = s8.shape[2]
            assert torch.equal(s8.view(n, nmv, tt, 4, 16, w).sum(3), s16), "sad16 != sum of its 8x8 blocks"
            assert torch.equal(s16.view(n, nmv, tt, 4, 4, w).sum(4), s32), "sad32 != sum of its 16x16 blocks"
            assert torch.equal(s32.sum(3, keepdim=True), s64), "sad64 != sum of its 32x32 blocks"


def _spot_checks(L, planes_host_lookup, g, cur, ref, pad, rng, ctus):
    wc = cur.shape[1] // 64
    for ctu in ctus:
        cy, cx = (ctu // wc) * 64, (ctu % wc) * 64
        for dy, dx in ((0, 0), (-rng, -rng), (rng, rng), (-rng, rng), (rng, -rng), (3, -2), (-17, 41)):
            d = np.abs(cur[cy:cy + 64, cx:cx + 64].astype(int)
                       - ref[pad + cy + dy:pad + cy + dy + 64, pad + cx + dx:pad + cx + dx + 64].astype(int))
            assert planes_host_lookup(3, L.xmi_sad_surface_index(C.byref(g), 3, ctu, 0, dy, dx)) == d.sum()
            assert planes_host_lookup(2, L.xmi_sad_surface_index(C.byref(g), 2, ctu, 3, dy, dx)) == d[32:, 32:].sum()
            assert planes_host_lookup(1, L.xmi_sad_surface_index(C.byref(g), 1, ctu, 6, dy, dx)) == d[16:32, 32:48].sum()
            assert planes_host_lookup(0, L.xmi_sad_surface_index(C.byref(g), 0, ctu, 8 * 5 + 2, dy, dx)) == d[40:48, 16:24].sum()


def test_surface_full_size_1080p_properties(L):
    """The 1080p benchmark configuration at full size (1920x1088, R = 64: 510 CTUs x 16 641 MVs x 85 PUs, 1.5 GB of output), device
    resident: the whole pyramid must be consistent, and direct SADs at the window corners and a few other MVs must match."""
    torch = pytest.importorskip("torch")
    rng, pad, w, h = 64, 80, 1920, 1088
    cur, ref = make_frames(w, h, pad, seed=7, kind="smooth")
    nctu = (w // 64) * (h // 64)
    g = capi.SadGeom(w, h, w, w + 2 * pad, rng)
    d_cur, d_ref = torch.from_numpy(cur).cuda(), torch.from_numpy(ref).cuda()
    dts = (torch.int16, torch.int16, torch.int32, torch.int32)
    planes = [torch.full((layout.plane_elems(nctu, rng, lvl),), -21846, dtype=dts[lvl], device="cuda") for lvl in range(4)]
    capi.check(L.xmi_sad_surface_dev(C.byref(g), d_cur.data_ptr(), d_ref.data_ptr() + pad * (w + 2 * pad) + pad,
                                     *[p.data_ptr() for p in planes], 0, nctu, None))
    torch.cuda.synchronize()
    upl = [planes[0].view(torch.uint16), planes[1].view(torch.uint16), planes[2], planes[3]]
    _pyramid_checks(torch, upl, nctu, rng, 0, nctu)
    _spot_checks(L, lambda lvl, idx: int(upl[lvl][idx].item()), g, cur, ref, pad, rng, (0, 29, 255, 480, 509))


def test_surface_2160p_range128_full_frame(L):
    """The 2160p configuration at full size (3840x2176, R = 128: 2 040 CTUs x 66 049 MVs x 85 PUs, 24.3 GB of output in one
    launch), device resident: the pyramid must be consistent over every CTU, and direct SADs at the window corners and a few
    other MVs must match in the four corner CTUs, on all four picture edges and inside."""
    torch = pytest.importorskip("torch")
    rng, pad, w, h = 128, 144, 3840, 2176
    cur, ref = make_frames(w, h, pad, seed=9, kind="smooth")
    wc, hc = w // 64, h // 64
    nctu = wc * hc
    g = capi.SadGeom(w, h, w, w + 2 * pad, rng)
    d_cur, d_ref = torch.from_numpy(cur).cuda(), torch.from_numpy(ref).cuda()
    dts = (torch.int16, torch.int16, torch.int32, torch.int32)
    planes = [torch.full((layout.plane_elems(nctu, rng, lvl),), -21846, dtype=dts[lvl], device="cuda") for lvl in range(4)]
    capi.check(L.xmi_sad_surface_dev(C.byref(g), d_cur.data_ptr(), d_ref.data_ptr() + pad * (w + 2 * pad) + pad,
                                     *[p.data_ptr() for p in planes], 0, nctu, None))
    torch.cuda.synchronize()
    upl = [planes[0].view(torch.uint16), planes[1].view(torch.uint16), planes[2], planes[3]]
    _pyramid_checks(torch, upl, nctu, rng, 0, nctu)
    ctus = (0, wc - 1, nctu - wc, nctu - 1,                      # corners
            wc // 2, (hc // 2) * wc, (hc // 2) * wc + wc - 1, nctu - wc // 2,   # top, left, right, bottom edge
            (hc // 2) * wc + wc // 2, 1000)                      # inside
    _spot_checks(L, lambda lvl, idx: int(upl[lvl][idx].item()), g, cur, ref, pad, rng, ctus)


def test_host_form_from_two_threads(L, orc):
    """xmi_sad_surface stages frames and planes in library-owned buffers: two host threads calling it at once, with
    different picture sizes (so that one call's buffers would have to grow under the other), both get the oracle's surface."""
    import threading
    jobs = [(128, 64, 32, 11), (64, 64, 16, 12), (64, 128, 32, 13), (128, 128, 16, 14)]
    out, err = {}, []

    def work(k):
        try:
            for rep in range(3):
                for j in jobs[k::2]:
                    w, h, rng, seed = j
                    pad = ((rng + 4 + 15) // 16) * 16
                    cur, ref = make_frames(w, h, pad, seed=seed)
                    out[(j, rep)] = (gpu_surface(L, cur, ref, pad, rng), cur, ref, pad, rng)
        except Exception as e:      # noqa: BLE001 - reported by the parent
            err.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    assert len(out) == 12
    for (j, rep), (got, cur, ref, pad, rng) in out.items():
        want = O.sad_surface(orc, cur, ref, pad, rng)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b, err_msg=f"{j} rep {rep}")
