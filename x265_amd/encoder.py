"""Python handle of the encoder in libx265_amd.so (xmi_encoder_*, include/x265_amd.h) and the picture preparation of the
reference's command line, for the tests and the benchmark. Nothing is computed here."""
import ctypes as C

import numpy as np

from .capi import SLICE_I, SLICE_P, XmiError, load


def align_frame(y, u, v, width, height):
    """What the reference's command line hands its encoder when the picture size is no multiple of 64 (xFrameAligning,
    src/preProcess.cpp:33): columns are padded with the row's last sample; every added row is a copy of `width` bytes of
    the input buffer from the start of the plane's last row on - the row itself and, past its end, whatever follows it
    in the Y, U, V buffer (zeros after V). -> contiguous planes of width x height, width/2 x height/2."""
    flat = np.concatenate([y.ravel(), u.ravel(), v.ravel(), np.zeros(width, np.uint8)])
    out, start = [], 0
    for p, s in ((y, 1), (u, 2), (v, 2)):
        ph, pw, aw, ah = p.shape[0], p.shape[1], width // s, height // s
        a = np.zeros((ah, aw), np.uint8)
        a[:ph, :pw] = p
        a[:ph, pw:] = p[:, -1:]
        if ah > ph:
            last = start + (ph - 1) * pw
            a[ph:, :] = flat[last:last + aw]
        out.append(np.ascontiguousarray(a))
        start += ph * pw
    return out


class Encoder:
    """one encoder instance; pictures are (y, u, v) uint8 arrays of the display size given to the constructor"""

    def __init__(self, width, height, qp, strong_intra_smoothing=False, lanes=None, spin_us=0, speculate=None, workers=None, rdoq=False, picture_hash=True):
        self.lib = load()
        self.width, self.height = width, height
        self.W, self.H = (width + 63) // 64 * 64, (height + 63) // 64 * 64
        self.lib.xmi_encoder_create.restype = C.c_void_p
        self.lib.xmi_encoder_create.argtypes = [C.c_int] * 4
        self.lib.xmi_encoder_destroy.argtypes = [C.c_void_p]
        self.lib.xmi_encoder_configure.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.lib.xmi_encoder_encode.restype = C.c_int64
        self.lib.xmi_encoder_encode.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_size_t]
        self.lib.xmi_encoder_reconstruction.argtypes = [C.c_void_p] * 4
        self.lib.xmi_encoder_timing.argtypes = [C.c_void_p] * 2
        self.lib.xmi_last_error.restype = C.c_char_p
        self.handle = self.lib.xmi_encoder_create(self.W, self.H, qp, int(bool(strong_intra_smoothing)))
        if not self.handle:
            raise XmiError(self.lib.xmi_last_error().decode())
        if lanes is not None or spin_us:
            rc = self.lib.xmi_encoder_configure(self.handle, lanes or 1, spin_us)
            if rc != 0:
                raise XmiError(self.lib.xmi_last_error().decode())
        if workers is not None:
            self.lib.xmi_encoder_workers.argtypes = [C.c_void_p, C.c_int]
            if self.lib.xmi_encoder_workers(self.handle, workers) != 0:
                raise XmiError(self.lib.xmi_last_error().decode())
        if speculate is not None:
            self.lib.xmi_encoder_speculate.argtypes = [C.c_void_p, C.c_int]
            if self.lib.xmi_encoder_speculate(self.handle, speculate) != 0:
                raise XmiError(self.lib.xmi_last_error().decode())
        if rdoq:
            self.lib.xmi_encoder_rdoq.argtypes = [C.c_void_p, C.c_int]
            if self.lib.xmi_encoder_rdoq(self.handle, 1) != 0:
                raise XmiError(self.lib.xmi_last_error().decode())
        if not picture_hash:
            self.lib.xmi_encoder_picture_hash.argtypes = [C.c_void_p, C.c_int]
            if self.lib.xmi_encoder_picture_hash(self.handle, 0) != 0:
                raise XmiError(self.lib.xmi_last_error().decode())
        self.out = np.zeros(self.W * self.H * 3 + 4096, np.uint8)

    def close(self):
        if self.handle:
            self.lib.xmi_encoder_destroy(self.handle)
            self.handle = None

    def encode(self, y, u, v, intra):
        """-> the access unit's bytes"""
        if y.shape != (self.H, self.W):
            y, u, v = align_frame(y, u, v, self.W, self.H)
        else:
            y, u, v = (np.ascontiguousarray(p) for p in (y, u, v))
        n = self.lib.xmi_encoder_encode(self.handle, y.ctypes.data, u.ctypes.data, v.ctypes.data, SLICE_I if intra else SLICE_P, self.out.ctypes.data, self.out.size)
        if n <= 0:
            raise XmiError(f"xmi_encoder_encode returned {n}: {self.lib.xmi_last_error().decode()}")
        return bytes(self.out[:n])

    def submit(self, y, u, v, intra):
        """first step of encode(): everything but the picture hash; collect() hands out the access units in order"""
        if y.shape != (self.H, self.W):
            y, u, v = align_frame(y, u, v, self.W, self.H)
        else:
            y, u, v = (np.ascontiguousarray(p) for p in (y, u, v))
        self.lib.xmi_encoder_submit.argtypes = [C.c_void_p] * 4 + [C.c_int]
        rc = self.lib.xmi_encoder_submit(self.handle, y.ctypes.data, u.ctypes.data, v.ctypes.data, SLICE_I if intra else SLICE_P)
        if rc != 0:
            raise XmiError(f"xmi_encoder_submit returned {rc}: {self.lib.xmi_last_error().decode()}")

    def collect(self):
        """-> the bytes of the oldest access unit that waits"""
        self.lib.xmi_encoder_collect.restype = C.c_int64
        self.lib.xmi_encoder_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        n = self.lib.xmi_encoder_collect(self.handle, self.out.ctypes.data, self.out.size)
        if n <= 0:
            raise XmiError(f"xmi_encoder_collect returned {n}: {self.lib.xmi_last_error().decode()}")
        return bytes(self.out[:n])

    def reconstruction(self):
        y = np.zeros((self.H, self.W), np.uint8)
        u, v = np.zeros((self.H // 2, self.W // 2), np.uint8), np.zeros((self.H // 2, self.W // 2), np.uint8)
        self.lib.xmi_encoder_reconstruction(self.handle, y.ctypes.data, u.ctypes.data, v.ctypes.data)
        return y, u, v

    def timing(self):
        t = (C.c_uint64 * 10)()
        self.lib.xmi_encoder_timing(self.handle, t)
        n = max(1, int(t[4]))
        return {"analysis_ms": t[0] / 1e6, "deblock_ms": t[1] / 1e6, "sao_ms": t[2] / 1e6, "write_ms": t[3] / 1e6, "round_trips": int(t[4]), "requests": int(t[5]),
                "us_per_round_trip": {"total": t[0] / 1e3 / n, "job_tables": t[6] / 1e3 / n, "launch_call": t[7] / 1e3 / n, "gpu_wait": t[8] / 1e3 / n,
                                      "read_back": t[9] / 1e3 / n}}


def scene_changed(enc, y_aligned, pictures_done, intra_period, rec_y):
    """sceneChangeDetection (src/preProcess.cpp:104-152) with the library's SAD: 32x32 blocks of the picture about to be coded
    against a reconstructed picture; a block with SAD above 320 * 16 is a motion block, and 85 % motion blocks make a scene
    change. As the reference computes it: its loops over the CTUs never move the two pointers, so it is the four blocks of the
    first CTU, each counted once per CTU of the picture; and `rec_y` is the reconstruction of the picture BEFORE the last one -
    the function reads the slot h->pFrameRec after xEncodeFrame has exchanged the two pictures (src/encode.cpp:6087-6092), which
    then holds the last picture's reference. A cut at picture c that falls into the first SCENE_CHANGE_INTERVAL pictures after
    an I picture is therefore still seen at picture c + 1."""
    if pictures_done <= 0:
        return True
    if intra_period > 1 and pictures_done % intra_period == 0:
        return False
    from . import capi
    L = capi.load()
    ctus = (enc.W // 64) * (enc.H // 64)
    threshold = int(0.85 * (ctus * 4) + 0.5 + 1e-6)
    motion = 0
    for oy, ox in ((0, 0), (0, 32), (32, 0), (32, 32)):
        a, b = y_aligned[oy:, ox:], rec_y[oy:, ox:]
        sad = L.xmi_sad(32, a.ctypes.data, enc.W, b.ctypes.data, enc.W)      # xSadN[3]: 32x32
        if sad < 0:
            raise XmiError(L.xmi_last_error().decode())
        motion += ctus * int(sad > 320 * 16)
    return motion >= threshold


def encode_clip(frames, width, height, qp, intra_period=100, scene_change=False, **kw):
    """frames: iterable of (y, u, v); slice types as the reference's command line picks them (src/x265.cpp:236-252: picture i is
    an I slice when i % intra_period == 0 - or, with scene_change (its -sceneChange switch, on by default there), when more than
    SCENE_CHANGE_INTERVAL = 4 P pictures have passed and the detection fires - and the picture before it was none)
    -> stream bytes"""
    enc = Encoder(width, height, qp, **kw)
    try:
        out, prev_intra, interval = [], False, 0
        recs = []      # luma reconstructions of the last two pictures (only kept for the detection)
        for i, (y, u, v) in enumerate(frames):
            if y.shape != (enc.H, enc.W):
                y, u, v = align_frame(y, u, v, enc.W, enc.H)
            # more than SCENE_CHANGE_INTERVAL = 4 P pictures since the last I picture: at least five pictures are coded
            changed = scene_change and interval > 4 and scene_changed(enc, np.ascontiguousarray(y), i, intra_period, recs[-2])
            intra = (i % intra_period == 0 or changed) and not prev_intra
            interval = 0 if intra else interval + 1
            out.append(enc.encode(y, u, v, intra))
            if scene_change:
                recs = recs[-1:] + [enc.reconstruction()[0]]
            prev_intra = intra
        return b"".join(out)
    finally:
        enc.close()
