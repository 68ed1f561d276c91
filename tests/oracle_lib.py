"""ctypes binding of oracle/libx265_oracle.so - the CPU checker. Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libx265_oracle.so")

SLICE_P, SLICE_I, MODE_INVALID = 1, 2, 255


class OrcMV(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16)]


class OrcMEResult(C.Structure):
    _fields_ = [("mv", OrcMV), ("best_cost", C.c_uint32), ("best_sad", C.c_uint32),
                ("mvp_idx", C.c_int), ("n_int_sads", C.c_int), ("n_sub_sads", C.c_int)]


class OrcChromaResult(C.Structure):
    _fields_ = [("best_idx", C.c_int), ("best_mode", C.c_int), ("modes", C.c_int * 5), ("satd_u", C.c_uint32 * 5), ("satd_v", C.c_uint32 * 5)]


class OrcDeblockParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "stride_y", "stride_c", "tc_offset", "beta_offset", "cb_qp_offset", "cr_qp_offset")]


class OrcIntraResult(C.Structure):
    _fields_ = [("best_mode", C.c_int), ("pred_idx", C.c_int), ("best_cost", C.c_double), ("satd", C.c_uint32 * 35)]


def _build():
    src = os.path.join(ROOT, "oracle", "x265_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libx265_oracle.so"],
                              stdout=subprocess.DEVNULL)


def load():
    _build()
    lib = C.CDLL(_SO)
    vp, i, u = C.c_void_p, C.c_int, C.c_uint
    for name in ("orc_sad", "orc_ssd", "orc_satd_hm", "orc_satd_x264", "orc_satd_table"):
        f = getattr(lib, name)
        f.restype = C.c_uint32
        f.argtypes = [i, vp, i, vp, i]
    lib.orc_sad_wh.restype = C.c_uint32
    lib.orc_sad_wh.argtypes = [i, i, vp, i, vp, i]
    lib.orc_sub_dct.restype = None
    lib.orc_sub_dct.argtypes = [vp, vp, i, vp, i, i, u]
    lib.orc_quant.restype = C.c_uint32
    lib.orc_quant.argtypes = [vp, vp, i, u, i, i, i]
    lib.orc_dequant.restype = None
    lib.orc_dequant.argtypes = [vp, vp, i, u, i, i]
    lib.orc_idct_add.restype = None
    lib.orc_idct_add.argtypes = [vp, i, vp, i, vp, i, u]
    lib.orc_tu_chain.restype = C.c_uint32
    lib.orc_tu_chain.argtypes = [vp, vp, vp, vp, i, i, i, u, u, i]
    lib.orc_sad_surface.restype = None
    lib.orc_sad_surface.argtypes = [vp, i, vp, i, i, i, i, i, vp, vp, vp, vp, i, i]
    lib.orc_mv_component_bits.restype = u
    lib.orc_mv_component_bits.argtypes = [i]
    lib.orc_pred_inter_luma.restype = None
    lib.orc_pred_inter_luma.argtypes = [vp, vp, i, i, i, i, i]
    lib.orc_sao_apply_plane.restype = None
    lib.orc_sao_apply_plane.argtypes = [vp, i, vp, i, i, i, i, vp]
    lib.orc_sao_stats_ctu.restype = None
    lib.orc_sao_stats_ctu.argtypes = [vp, i, vp, i, i, i, i, i, i, i, vp, vp]
    lib.orc_fill_mvp_cand.restype = i
    lib.orc_fill_mvp_cand.argtypes = [i, i, vp, vp, i, i, i, i, i, i, vp]
    lib.orc_intra_chroma_search.restype = None
    lib.orc_intra_chroma_search.argtypes = [vp, vp, i, vp, vp, i, i, C.POINTER(OrcChromaResult)]
    lib.orc_deblock_strengths.restype = None
    lib.orc_deblock_strengths.argtypes = [i, i, i, vp, vp, vp, vp, vp]
    lib.orc_deblock_frame.restype = None
    lib.orc_deblock_frame.argtypes = [vp, vp, vp, C.POINTER(OrcDeblockParams), vp, vp, vp]
    lib.orc_pred_inter_chroma.restype = None
    lib.orc_pred_inter_chroma.argtypes = [vp, vp, i, i, i, i, i]
    lib.orc_motion_search.restype = None
    lib.orc_motion_search.argtypes = [vp, vp, i, i, C.POINTER(OrcMV), i, OrcMV, OrcMV, C.c_double, i,
                                      C.POINTER(OrcMEResult)]
    lib.orc_padding_ref.restype = None
    lib.orc_padding_ref.argtypes = [vp, i, u]
    lib.orc_filter_ref.restype = None
    lib.orc_filter_ref.argtypes = [vp, vp, i, i]
    lib.orc_intra_pred.restype = None
    lib.orc_intra_pred.argtypes = [vp, vp, i, i, i, i]
    lib.orc_intra_search.restype = None
    lib.orc_intra_search.argtypes = [vp, i, vp, i, u, i, vp, C.c_double, C.POINTER(OrcIntraResult)]
    return lib


def p(a):
    """pointer to a C-contiguous array; the pointer object keeps the array alive (temporaries may be passed)"""
    assert a.flags["C_CONTIGUOUS"]
    ptr = C.c_void_p(a.ctypes.data)
    ptr._keep = a
    return ptr


def at(a, *idx):
    """pointer to element a[idx] of a C-contiguous array"""
    off = sum(int(i) * int(s) for i, s in zip(idx, a.strides))
    ptr = C.c_void_p(a.ctypes.data + off)
    ptr._keep = a
    return ptr


def sad_surface(lib, cur, ref_padded, pad, rng, pitch=None, ctu_begin=0, ctu_end=None):
    """cur: (H,W) u8; ref_padded: (H+2*pad, W+2*pad) u8. Returns (sad8, sad16, sad32, sad64)."""
    h, w = cur.shape
    nctu = (h // 64) * (w // 64)
    nmv = 2 * rng + 1
    pitch = pitch or nmv
    if ctu_end is None:
        ctu_end = nctu
    s8 = np.zeros((nctu, 64, nmv, pitch), np.uint16)
    s16 = np.zeros((nctu, 16, nmv, pitch), np.uint16)
    s32 = np.zeros((nctu, 4, nmv, pitch), np.uint32)
    s64 = np.zeros((nctu, 1, nmv, pitch), np.uint32)
    lib.orc_sad_surface(p(cur), cur.strides[0], at(ref_padded, pad, pad), ref_padded.strides[0],
                        w, h, rng, pitch, p(s8), p(s16), p(s32), p(s64), ctu_begin, ctu_end)
    return s8, s16, s32, s64


def sao_params(table):
    """(CTUs, >= 6) int array of typeIdx, subTypeIdx, offset[4] -> packed 8-byte records of the C structs"""
    out = np.zeros((len(table), 8), np.uint8)
    out[:, 0] = table[:, 0].astype(np.int8).view(np.uint8)
    out[:, 1] = table[:, 1].astype(np.uint8)
    out[:, 2:6] = table[:, 2:6].astype(np.int8).view(np.uint8)
    return np.ascontiguousarray(out)
