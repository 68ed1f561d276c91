"""GPU: xmi_deblock_strengths_static, the entry point that allocates nothing (its staging lives in an array of the
library's device image), against xmi_deblock_strengths on the same maps, bit for bit, through the C ABI: two calls in a
row, the second on a smaller picture, so that what the first call left in the array would show if anything of it were read."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O
from x265_amd import capi

pytestmark = pytest.mark.gpu


def free_device_bytes():
    """free bytes of the current device as the HIP runtime of this process reports them (hipMemGetInfo)"""
    import torch
    return torch.cuda.mem_get_info()[0]


def random_maps(w, h, intra, seed):
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
            mv[y0 // 8:(y0 + size) // 8, x0 // 8:(x0 + size) // 8] = rng.integers(-40, 41, 2)
    for y0 in range(0, h, 64):
        for x0 in range(0, w, 64):
            split(x0, y0, 64)
    cbf = (rng.random(lg.shape) < 0.3).astype(np.uint8)
    return lg, cbf, mv


def both(L, w, h, intra, seed):
    lg, cbf, mv = random_maps(w, h, intra, seed)
    n = (w // 8) * (h // 4)
    want_v, want_h = np.full(n, 9, np.uint8), np.full(n, 9, np.uint8)
    capi.check(L.xmi_deblock_strengths(w, h, intra, O.p(lg), O.p(cbf), O.p(mv), O.p(want_v), O.p(want_h)))
    got_v, got_h = np.full(n, 7, np.uint8), np.full(n, 7, np.uint8)
    free_before = free_device_bytes()
    capi.check(L.xmi_deblock_strengths_static(w, h, intra, O.p(lg), O.p(cbf), None if intra else O.p(mv), O.p(got_v), O.p(got_h)))
    assert free_device_bytes() == free_before, "the call changed the amount of free device memory"
    np.testing.assert_array_equal(got_v, want_v)
    np.testing.assert_array_equal(got_h, want_h)
    # and against the CPU checker's restatement of the reference's per-unit loops
    orc = O.load()
    orc_v, orc_h = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    orc.orc_deblock_strengths(w, h, intra, O.p(lg), O.p(cbf), O.p(mv), O.p(orc_v), O.p(orc_h))
    np.testing.assert_array_equal(got_v, orc_v)
    np.testing.assert_array_equal(got_h, orc_h)
    assert set(np.unique(got_v)) <= {0, 1, 2} and (got_v != 0).any()
    return got_v, got_h


def test_static_staging_matches_allocating_entry_point_twice():
    L = capi.load()
    capi.check(L.xmi_init(0))
    both(L, 1920, 1088, 0, 1)           # P picture, fills most of the smaller offsets of the array
    both(L, 448, 256, 0, 2)             # smaller picture right after: offsets move, stale bytes lie behind every region
    both(L, 200, 72, 1, 3)              # I slice (no vectors copied), size off the CTU grid
    both(L, 3840, 2176, 0, 4)           # the 2160p geometry, close to the limit


def test_static_staging_refuses_what_it_cannot_hold():
    L = capi.load()
    capi.check(L.xmi_init(0))
    z = np.zeros(8, np.uint8)
    assert L.xmi_deblock_strengths_static(8192, 4320, 1, O.p(z), O.p(z), None, O.p(z), O.p(z)) == -1
    assert b"4096x2304" in L.xmi_last_error()
    assert L.xmi_deblock_strengths_static(1920, 1084, 1, O.p(z), O.p(z), None, O.p(z), O.p(z)) == -1
