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
