"""GPU: xmi_analysis_p_picture (the P-picture analysis loop of libx265_amd.so: host decision logic + GPU evaluator) through
the C ABI against what the reference's CTU loop decided in bit-identical encodes - decisions, vectors, levels, RD costs,
estimation contexts, the unfiltered reconstruction and the state carried between pictures: the committed pictures of
tests/golden/ctu_trace.npz, and freshly recorded clips (the reference's trace encoder, oracle/_ref/x265_ctu_trace, runs on
the host of the GPU box) at 416x240 and at 1920x1080, all P pictures in order with the cost statistics kept by the loop."""
import pytest

from tests import ctu_trace as T
from tests.prereq import needs_ref
from tests.test_analysis_cpu import check_fixture_picture
from x265_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = capi.load()
    capi.check(lib.xmi_init(0))
    return lib


@pytest.mark.parametrize("tag", ["416x240_q32_p1", "416x240_q32_p2", "416x240_q32_p7", "416x240_q22_p1"])
def test_recorded_pictures_are_reproduced(L, tag):
    out = check_fixture_picture(lambda w, h: T.Runner(L, "xmi_analysis_", w, h), tag)
    assert out["counters"][0] > 0


@pytest.mark.parametrize("lanes,spin,tag", [(1, 200, "416x240_q32_p2"), (2, 0, "416x240_q32_p2"), (3, 20, "416x240_q22_p1"), (4, 200, "416x240_q32_p7")])
def test_row_lanes_and_polling_do_not_change_the_result(L, lanes, spin, tag):
    check_fixture_picture(lambda w, h: T.Runner(L, "xmi_analysis_", w, h, lanes, spin), tag)


@needs_ref("x265_ctu_trace", how="make -C oracle ctu_trace")
@pytest.mark.parametrize("w,h,frames,qp,seed,lanes", [(416, 240, 8, 32, 1234, None), (416, 240, 4, 22, 1234, 2), (1920, 1080, 3, 32, 7, None),
                                                      (1920, 1080, 3, 27, 7, 3), (1920, 1080, 2, 22, 7, 1), (1920, 1080, 2, 37, 7, 4)])
def test_fresh_trace_with_carried_statistics(L, tmp_path, w, h, frames, qp, seed, lanes):
    pics, _ = T.record_trace(str(tmp_path), w, h, frames, qp, seed)
    r = T.Runner(L, "xmi_analysis_", pics[0]["w"], pics[0]["h"], lanes, 200 if lanes in (None, 1) else 0)
    try:
        r.set_state(pics[0], pics[1])
        for k in range(1, len(pics)):
            pic = pics[k]
            assert pic["slice"] == 1
            st = r.get_state(pic["rows"], pic["cols"])
            st["frac_start"] = pic["frac_start"]
            r.set_state(st, st)
            out = r.run(pic)
            assert T.compare(pic, out, r.get_state(pic["rows"], pic["cols"])) == [], f"picture {k}"
    finally:
        r.close()


def test_bad_arguments_are_refused(L):
    import ctypes as C
    L.xmi_analysis_create.restype = C.c_void_p
    L.xmi_analysis_create.argtypes = [C.c_int, C.c_int]
    assert not L.xmi_analysis_create(416, 240)
    assert b"multiple of 64" in L.xmi_last_error()
    a = L.xmi_analysis_create(128, 128)
    assert a
    L.xmi_analysis_configure.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert L.xmi_analysis_configure(a, 0, 0) != 0 and L.xmi_analysis_configure(a, 9, 0) != 0 and L.xmi_analysis_configure(a, 1, -1) != 0
    assert L.xmi_analysis_configure(a, 2, 50) == 0
    L.xmi_analysis_destroy.argtypes = [C.c_void_p]
    L.xmi_analysis_destroy(a)
