"""CPU: the built libraries export every entry point the public header declares (no compute calls)."""
import os
import re
import subprocess

import pytest

from tests.prereq import needs_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "x265_amd", "lib", "libx265_amd.so")
DROPIN = os.path.join(ROOT, "x265_amd", "lib", "libx265_amd_dropin.so")


def exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def declared():
    text = open(os.path.join(ROOT, "include", "x265_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(xmi_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported():
    decl = declared()
    assert len(decl) >= 20
    missing = decl - exported(LIB)
    assert not missing, f"declared in include/x265_amd.h but not exported: {sorted(missing)}"


def test_library_loads_and_reports_without_gpu():
    import ctypes as C
    lib = C.CDLL(LIB)
    lib.xmi_version.restype = C.c_char_p
    assert b"gfx950" in lib.xmi_version()
    lib.xmi_last_error.restype = C.c_char_p
    assert isinstance(lib.xmi_last_error(), bytes)
    lib.xmi_device_count.restype = C.c_int
    assert lib.xmi_device_count() >= 0


def test_dropin_exports_the_reference_primitive_interface():
    # mangled names are the ones an object compiled from the reference's encode.cpp imports
    want = {"xSadN", "xSad_AMP", "xSatdN", "xSsdN", "xSatd_AMP", "_Z7xSubDctPsPhjS0_jS_S_ij", "_Z6xQuantPsS_jjii10xSliceType",
            "_Z8xDeQuantPsS_jjii", "_Z8xIDctAddPhjPsjS_S0_S0_ij", "xmi_install_primitives",
            "_Z17xPredInterLumaBlkPKhPhjjj3xMVPs", "_Z19xPredInterChromaBlkPKhPhjjj3xMVPs", "_Z13xEncIntraPredPhS_jjjj",
            "_Z12xFillMvpCandP3xMVPKhPKS_jjjj"}
    assert want <= exported(DROPIN)


def test_no_cpu_fallback_when_no_gpu():
    """Without a GPU the compute entry points must fail loudly, never compute on the host."""
    import ctypes as C
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    lib = C.CDLL(LIB)
    lib.xmi_device_count.restype = C.c_int
    if lib.xmi_device_count() > 0:
        pytest.skip("a GPU is visible")
    lib.xmi_last_error.restype = C.c_char_p
    import numpy as np
    a = np.zeros((8, 8), np.uint8)
    lib.xmi_sad.restype = C.c_int64
    lib.xmi_sad.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    assert lib.xmi_sad(8, a.ctypes.data, 8, a.ctypes.data, 8) == -3
    assert b"no CPU fallback" in lib.xmi_last_error()


GPU_ALL = os.path.join(ROOT, "oracle", "_ref", "x265_gpu_all")


@needs_ref("x265_gpu_all", how="make -C oracle trace")
def test_reference_library_binds_to_the_dropin():
    """In the all-GPU encoder the reference's own shared library must take every pixel primitive and prediction
    function from libx265_amd_dropin.so (symbol interposition by link order): read the dynamic linker's binding log."""
    env = dict(os.environ, LD_BIND_NOW="1", LD_DEBUG="bindings")
    log = subprocess.run([GPU_ALL], env=env, capture_output=True, text=True, timeout=60).stderr
    want = {"xSadN", "xSad_AMP", "xSatdN", "xSsdN", "_Z7xSubDctPsPhjS0_jS_S_ij", "_Z6xQuantPsS_jjii10xSliceType", "_Z8xDeQuantPsS_jjii",
            "_Z8xIDctAddPhjPsjS_S0_S0_ij", "_Z17xPredInterLumaBlkPKhPhjjj3xMVPs", "_Z19xPredInterChromaBlkPKhPhjjj3xMVPs",
            "_Z13xEncIntraPredPhS_jjjj", "_Z12xFillMvpCandP3xMVPKhPKS_jjjj"}
    bound = {}
    for line in log.splitlines():
        m = re.search(r"binding file \S*libx265ref_me\.so \[0\] to (\S+) \[0\]: normal symbol `([^']+)'", line)
        if m and m.group(2) in want:
            bound[m.group(2)] = os.path.basename(m.group(1))
    assert set(bound) == want, f"no binding seen for {sorted(want - set(bound))}"
    # xEncIntraPred goes to the executable first: its intra-search hook forwards to the next definition, the drop-in's
    allowed = {k: {"libx265_amd_dropin.so"} for k in want}
    allowed["_Z13xEncIntraPredPhS_jjjj"].add("x265_gpu_all")
    wrong = {k: v for k, v in bound.items() if v not in allowed[k]}
    assert not wrong, f"bound to the reference's own definitions: {wrong}"


def test_cu_mv_clip_reproduces_the_boxes_of_a_real_encode():
    """xmi_cu_mv_clip (host arithmetic, no device) against the clip boxes the reference's encoder handed to its motion
    searches (tests/golden/me_trace.npz: every request of three P pictures of the 448x256 coded clip carries its box)."""
    import ctypes as C
    import numpy as np
    from x265_amd import capi
    L = capi.load()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "me_trace.npz"))
    seen = set()
    for poc in g["pocs"]:
        for r in g[f"p{poc}_req"]:
            mn, mx = (C.c_int16 * 2)(), (C.c_int16 * 2)()
            assert L.xmi_cu_mv_clip(int(r[0]) // 64, int(r[1]) // 64, 7, 4, mn, mx) == 0
            assert (mn[0], mn[1], mx[0], mx[1]) == tuple(int(v) for v in r[8:12]), f"CTU {int(r[0]) // 64},{int(r[1]) // 64}"
            seen.add((int(r[0]) // 64, int(r[1]) // 64))
    assert len(seen) >= 10 and {c[0] for c in seen} >= {0, 6} and {c[1] for c in seen} >= {0, 3}      # both picture edges in both directions
    assert L.xmi_cu_mv_clip(7, 0, 7, 4, mn, mx) == -1
