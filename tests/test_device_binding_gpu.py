"""GPU: which device a call runs on. The library is bound to one device per process (xmi_init, or the first call); entry points
that take host pointers or a handle make that device current for whatever thread calls them, entry points that take the
caller's device pointers refuse a thread whose current device is another one."""
import ctypes as C
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import ctypes as C, sys, threading
import numpy as np
import torch
sys.path.insert(0, %r)
from x265_amd import capi
L = capi.load()
dev = int(sys.argv[1])
capi.check(L.xmi_init(dev))
a = np.arange(64, dtype=np.uint8).reshape(8, 8).copy(); b = np.zeros((8, 8), np.uint8)
out = {}
def work():                      # a fresh host thread starts on device 0
    out["sad"] = L.xmi_sad(8, a.ctypes.data, 8, b.ctypes.data, 8)
    out["dev"] = torch.cuda.current_device()
t = threading.Thread(target=work); t.start(); t.join()
assert out["sad"] == int(a.sum()), (out, L.xmi_last_error())
# a *_dev entry point from a thread on another device is refused, not moved
torch.cuda.set_device((dev + 1) %% torch.cuda.device_count())
x = torch.zeros(64, dtype=torch.uint8, device="cuda"); r = torch.zeros(1, dtype=torch.int32, device="cuda")
rc = L.xmi_block_metric_batch_dev(capi.METRIC_SAD, 8, 1, x.data_ptr(), x.data_ptr(), r.data_ptr(), None)
assert rc == -1, rc
print("ok", out)
"""


def test_host_pointer_entry_points_from_a_fresh_thread():
    from x265_amd import capi
    L = capi.load()
    capi.check(L.xmi_init(0))
    a = np.arange(256, dtype=np.uint8).reshape(16, 16).copy()
    b = np.zeros((16, 16), np.uint8)
    got = []
    th = [threading.Thread(target=lambda: got.append(L.xmi_sad(16, a.ctypes.data, 16, b.ctypes.data, 16))) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == [int(a.sum())] * 4


def test_bound_to_the_second_device_of_a_box_with_two():
    import os
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one device: a fresh thread's device 0 is the bound one anyway")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", CHILD % root, "1"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-1000:] + res.stderr[-2000:]
