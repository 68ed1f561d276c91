"""GPU: quarter-pel interpolation planes against the golden vectors made by the reference's compiled
xPredInterLumaBlk and against the oracle on random frames. Bit-exact."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from x265_amd import capi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def L():
    lib = capi.load()
    capi.check(lib.xmi_init(0))
    return lib


def planes_gpu(L, frame):
    rows, stride = frame.shape
    d = torch.from_numpy(np.ascontiguousarray(frame)).cuda()
    out = torch.zeros((16, rows, stride), dtype=torch.uint8, device="cuda")
    capi.check(L.xmi_interp_planes_dev(d.data_ptr(), stride, rows, stride, out.data_ptr(), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_planes_golden(L, golden_dir):
    g = np.load(os.path.join(golden_dir, "interp.npz"))
    oy, ox = g["origin"]
    for n in (8, 16, 32, 64):
        frames, mvs, want = g[f"ref{n}"], g[f"mv{n}"], g[f"pred{n}"]
        for fi in range(frames.shape[0]):
            pl = planes_gpu(L, frames[fi])
            for k, (mx, my) in enumerate(mvs):
                ph = (int(my) & 3) * 4 + (int(mx) & 3)
                y, x = oy + (int(my) >> 2), ox + (int(mx) >> 2)
                np.testing.assert_array_equal(pl[ph, y:y + n, x:x + n], want[fi, k], err_msg=f"n={n} frame {fi} mv {mx},{my}")


def test_planes_random_vs_oracle(L):
    orc = O.load()
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, (150, 212), dtype=np.uint8)     # ragged size: partial tiles on both axes
    frame[40:60, 50:90] = np.where(rng.random((20, 40)) < 0.5, 0, 255)
    pl = planes_gpu(L, frame)
    np.testing.assert_array_equal(pl[0, 4:-4, 4:-4], frame[4:-4, 4:-4])
    for _ in range(300):
        n = int(rng.choice([8, 16, 32, 64]))
        mx, my = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
        y0 = int(rng.integers(4 + 12, 150 - 4 - n - 12))
        x0 = int(rng.integers(4 + 12, 212 - 4 - n - 12))
        y, x = y0 + (my >> 2), x0 + (mx >> 2)
        if y < 4 or x < 4 or y + n > 146 or x + n > 208 or y < 7 or x < 7 or y + n > 142 or x + n > 204:
            continue
        dst = np.zeros((64, 64), np.uint8)
        orc.orc_pred_inter_luma(O.at(frame, y0, x0), O.p(dst), frame.strides[0], n, n, mx, my)
        np.testing.assert_array_equal(pl[(my & 3) * 4 + (mx & 3), y:y + n, x:x + n], dst[:n, :n])
