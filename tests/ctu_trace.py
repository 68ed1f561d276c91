"""Reader of the CTU traces written by oracle/ctu_trace_hook.cpp (record layout in that file's header), and the driver
that runs the P-picture analysis loop on one recorded picture and compares everything it leaves behind with the record.
Used by the CPU test (host logic + CPU evaluator, oracle/libx265_analysis_check.so) and by the GPU test (libx265_amd.so)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACE_EXE = os.path.join(ROOT, "oracle", "_ref", "x265_ctu_trace")


def read_trace(path):
    """-> list of pictures in coding order; picture-level arrays are assembled from the per-CTU records."""
    buf = np.fromfile(path, dtype=np.uint8)
    pos = 0

    def take(dtype, count):
        nonlocal pos
        n = np.dtype(dtype).itemsize * count
        out = buf[pos:pos + n].view(dtype)
        pos += n
        return out

    pics, ctus = [], []
    while pos < buf.size:
        tag = int(take(np.int32, 1)[0])
        if tag == 1:
            hdr = take(np.int32, 8)
            rec = {"poc": int(hdr[0]), "slice": int(hdr[1]), "x": int(hdr[2]), "y": int(hdr[3]), "qp": int(hdr[4]), "has_cost": int(hdr[5])}
            rec["rd_cost"] = float(take(np.float64, 1)[0])
            rec["frac"] = int(take(np.uint64, 1)[0])
            rec["ctx_in"] = take(np.uint8, 256)
            rec["ctx_out"] = take(np.uint8, 256)
            rec["info0"] = take(np.uint16, 256).reshape(16, 16)
            rec["info1"] = take(np.uint8, 256).reshape(16, 16)
            rec["mv"] = take(np.int16, 128).reshape(8, 8, 2)
            rec["mvd"] = take(np.int16, 128).reshape(8, 8, 2)
            rec["mvp_idx"] = take(np.uint8, 64).reshape(8, 8)
            rec["coef_y"] = take(np.int16, 4096).reshape(64, 64)
            rec["coef_u"] = take(np.int16, 1024).reshape(32, 32)
            rec["coef_v"] = take(np.int16, 1024).reshape(32, 32)
            ctus.append(rec)
        elif tag == 2:
            hdr = take(np.int32, 8)
            poc, slice_, w, h, qp, pad, rows, cols = (int(v) for v in hdr)
            pic = {"poc": poc, "slice": slice_, "w": w, "h": h, "qp": qp, "pad": pad, "rows": rows, "cols": cols}
            pc = pad // 2
            pic["src"] = (take(np.uint8, w * h).reshape(h, w), take(np.uint8, w * h // 4).reshape(h // 2, w // 2),
                          take(np.uint8, w * h // 4).reshape(h // 2, w // 2))
            pic["ref"] = (take(np.uint8, (w + 2 * pad) * (h + 2 * pad)).reshape(h + 2 * pad, w + 2 * pad),
                          take(np.uint8, (w // 2 + 2 * pc) * (h // 2 + 2 * pc)).reshape(h // 2 + 2 * pc, w // 2 + 2 * pc),
                          take(np.uint8, (w // 2 + 2 * pc) * (h // 2 + 2 * pc)).reshape(h // 2 + 2 * pc, w // 2 + 2 * pc))
            pic["rec"] = (take(np.uint8, w * h).reshape(h, w), take(np.uint8, w * h // 4).reshape(h // 2, w // 2),
                          take(np.uint8, w * h // 4).reshape(h // 2, w // 2))
            n = rows * cols * 5
            pic["sum"], pic["count"], pic["avg"] = take(np.float64, n).copy(), take(np.int32, n).copy(), take(np.float64, n).copy()
            pic["frac"] = take(np.uint64, rows).copy()
            pic["frac_start"] = take(np.uint64, rows).copy()
            assert len(ctus) == rows * cols and all(c["poc"] == poc for c in ctus)
            a = {"info0": np.zeros((h // 4, w // 4), np.uint16), "info1": np.zeros((h // 4, w // 4), np.uint8),
                 "mv": np.zeros((h // 8, w // 8, 2), np.int16), "mvd": np.zeros((h // 8, w // 8, 2), np.int16),
                 "mvp_idx": np.zeros((h // 8, w // 8), np.uint8), "coef_y": np.zeros((h, w), np.int16),
                 "coef_u": np.zeros((h // 2, w // 2), np.int16), "coef_v": np.zeros((h // 2, w // 2), np.int16),
                 "rd_cost": np.zeros((rows, cols), np.float64), "ctx_in": np.zeros((rows, cols, 256), np.uint8),
                 "ctx_out": np.zeros((rows, cols, 256), np.uint8)}
            for c in ctus:
                x, y = c["x"], c["y"]
                for k, g in (("info0", 16), ("info1", 16), ("mv", 8), ("mvd", 8), ("mvp_idx", 8), ("coef_y", 64), ("coef_u", 32), ("coef_v", 32)):
                    a[k][y * g:(y + 1) * g, x * g:(x + 1) * g] = c[k]
                a["rd_cost"][y, x] = c["rd_cost"]
                a["ctx_in"][y, x] = c["ctx_in"]
                a["ctx_out"][y, x] = c["ctx_out"]
            pic.update(a)
            pics.append(pic)
            ctus = []
        else:
            raise ValueError(f"bad tag {tag} at {pos}")
    return pics


def record_trace(tmp_dir, w, h, frames, qp, seed):
    """Encode a synthetic clip with the reference's encoder built with the trace hook; -> (pictures, stream bytes)."""
    yuv, out, trace = (os.path.join(tmp_dir, n) for n in ("in.yuv", "out.265", "ctu.bin"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_yuv.py"), str(w), str(h), str(frames), str(seed), yuv])
    env = dict(os.environ, XMI_CTU_TRACE=trace)
    subprocess.run([TRACE_EXE, "-i", yuv, "-o", out, "-w", str(w), "-h", str(h), "-f", str(frames), "-q", str(qp), "-ip", "100", "-sceneChange", "0",
                    "-t", "1"], check=True, capture_output=True, env=env, timeout=900)
    return read_trace(trace), open(out, "rb").read()


def p(a):
    return a.ctypes.data_as(C.c_void_p)


class Runner:
    """create / state / p_picture of either library, same argument lists (prefix orc_analysis_ or xmi_analysis_)."""

    def __init__(self, lib, prefix, w, h, lanes=None, spin_us=0, speculate=None, workers=None, defer=None):
        self.lib, self.w, self.h = lib, w, h
        self.create, self.destroy = getattr(lib, prefix + "create"), getattr(lib, prefix + "destroy")
        self.state, self.picture = getattr(lib, prefix + "state"), getattr(lib, prefix + "p_picture")
        self.create.restype = C.c_void_p
        self.create.argtypes = [C.c_int, C.c_int]
        self.destroy.argtypes = [C.c_void_p]
        self.state.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        self.picture.restype = C.c_int
        self.picture.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int, C.c_int] + [C.c_void_p] * 15
        self.handle = self.create(w, h)
        assert self.handle
        if lanes is not None or spin_us:
            conf = getattr(lib, prefix + "configure")
            conf.restype = C.c_int
            conf.argtypes = [C.c_void_p, C.c_int, C.c_int]
            rc = conf(self.handle, lanes or 1, spin_us)
            assert rc == 0, f"configure({lanes}, {spin_us}) returned {rc}"
        # How far a request reaches ahead is a switch of the CPU harness only (the library always runs at 2). On the CPU every
        # request costs its full pixel work, so the harness runs at 1 unless a test asks for more.
        if workers is not None:      # (CPU harness only) threads the CTU rows share as fibers
            wk = getattr(lib, prefix + "workers")
            wk.restype = C.c_int
            wk.argtypes = [C.c_void_p, C.c_int]
            assert wk(self.handle, workers) == 0
        if defer is not None:        # (CPU harness only) results held back until the rows ask for them; (on, n-th request fails)
            df = getattr(lib, prefix + "defer")
            df.restype = C.c_int
            df.argtypes = [C.c_void_p, C.c_int, C.c_int]
            assert df(self.handle, *defer) == 0
        if prefix.startswith("orc_") or speculate is not None:
            spec = getattr(lib, prefix + "speculate")
            spec.restype = C.c_int
            spec.argtypes = [C.c_void_p, C.c_int]
            assert spec(self.handle, 1 if speculate is None else speculate) == 0

    def close(self):
        self.destroy(self.handle)

    def set_state(self, pic_before, pic):
        """cost statistics as the previous picture left them, rate counters as the picture found them"""
        s, c, a = (np.ascontiguousarray(pic_before[k]) for k in ("sum", "count", "avg"))
        f = np.ascontiguousarray(pic["frac_start"])
        self.state(self.handle, 1, p(s), p(c), p(a), p(f))

    def get_state(self, rows, cols):
        n = rows * cols * 5
        s, c, a, f = np.zeros(n), np.zeros(n, np.int32), np.zeros(n), np.zeros(rows, np.uint64)
        self.state(self.handle, 0, p(s), p(c), p(a), p(f))
        return {"sum": s, "count": c, "avg": a, "frac": f}

    def run(self, pic):
        w, h, rows, cols = pic["w"], pic["h"], pic["rows"], pic["cols"]
        out = {"info0": np.zeros((h // 4, w // 4), np.uint16), "info1": np.zeros((h // 4, w // 4), np.uint8),
               "mv": np.zeros((h // 8, w // 8, 2), np.int16), "mvd": np.zeros((h // 8, w // 8, 2), np.int16),
               "mvp_idx": np.zeros((h // 8, w // 8), np.uint8), "coef_y": np.zeros((h, w), np.int16),
               "coef_u": np.zeros((h // 2, w // 2), np.int16), "coef_v": np.zeros((h // 2, w // 2), np.int16),
               "rec_y": np.zeros((h, w), np.uint8), "rec_u": np.zeros((h // 2, w // 2), np.uint8), "rec_v": np.zeros((h // 2, w // 2), np.uint8),
               "rd_cost": np.zeros((rows, cols), np.uint32), "ctx_in": np.zeros((rows, cols, 256), np.uint8),
               "ctx_out": np.zeros((rows, cols, 256), np.uint8), "counters": np.zeros(3, np.uint64)}
        src = [np.ascontiguousarray(a) for a in pic["src"]]
        ref = [np.ascontiguousarray(a) for a in pic["ref"]]
        rc = self.picture(self.handle, *[p(a) for a in src], *[p(a) for a in ref], pic["pad"], pic["qp"],
                          *[p(out[k]) for k in ("info0", "info1", "mv", "mvd", "mvp_idx", "coef_y", "coef_u", "coef_v", "rec_y", "rec_u", "rec_v",
                                                "rd_cost", "ctx_in", "ctx_out", "counters")])
        if getattr(self, "expect_rc", 0):
            assert rc == self.expect_rc, f"p_picture returned {rc}, expected {self.expect_rc}"
            return None
        assert rc == 0, f"p_picture returned {rc}"
        return out


def first_ctu_difference(got, want, key, grain, cols):
    d = np.argwhere(got != want)
    if d.size == 0:
        return None
    y, x = int(d[0][0]) // grain, int(d[0][1]) // grain
    return f"{key}: first difference in CTU ({x},{y}) of {cols} columns, {len(d)} elements differ"


def compare(pic, out, state_after):
    """every array the picture leaves behind, CTU decisions first so that a failure names the earliest stage"""
    msgs = []
    for key, grain in (("ctx_in", 1), ("rd_cost", 1), ("ctx_out", 1), ("mv", 8), ("info0", 16), ("info1", 16), ("mvd", 8), ("mvp_idx", 8),
                       ("coef_y", 64), ("coef_u", 32), ("coef_v", 32)):
        want = pic[key].astype(out[key].dtype) if key == "rd_cost" else pic[key]
        m = first_ctu_difference(out[key], want, key, grain, pic["cols"])
        if m:
            msgs.append(m)
    for i, key in enumerate(("rec_y", "rec_u", "rec_v")):
        m = first_ctu_difference(out[key], pic["rec"][i], key, 64 if i == 0 else 32, pic["cols"])
        if m:
            msgs.append(m)
    for key in ("sum", "count", "avg", "frac"):
        if not np.array_equal(state_after[key], pic[key]):
            msgs.append(f"carried state {key} differs")
    return msgs
