"""ctypes binding of libx265_amd.so (include/x265_amd.h). No computation happens in Python."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# X265_AMD_LIB lets a profiling run point at an experimental build of the same library
LIB_PATH = os.environ.get("X265_AMD_LIB") or os.path.join(_HERE, "lib", "libx265_amd.so")

XMI_OK = 0
SLICE_P, SLICE_I, MODE_INVALID = 1, 2, 255
METRIC_SAD, METRIC_SSD, METRIC_SATD_HM, METRIC_SATD_X264, METRIC_SATD_TAB = range(5)


class XmiError(RuntimeError):
    pass


class SadGeom(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("cur_stride", C.c_int), ("ref_stride", C.c_int),
                ("range", C.c_int)]


class MeRequest(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("size", C.c_int32), ("mvp", (C.c_int16 * 2) * 2),
                ("zero_mv_tested", C.c_int32), ("mv_min_ipel", C.c_int16 * 2), ("mv_max_ipel", C.c_int16 * 2),
                ("sqrt_lambda", C.c_uint32)]


class MeResult(C.Structure):
    _fields_ = [("mv", C.c_int16 * 2), ("best_cost", C.c_uint32), ("best_sad", C.c_uint32), ("mvp_idx", C.c_int32),
                ("n_int_sads", C.c_int32), ("n_sub_sads", C.c_int32)]


class IntraRequest(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("size", C.c_uint8), ("valids", C.c_uint8),
                ("strong_smoothing", C.c_uint8), ("mpm", C.c_uint8 * 3), ("reserved", C.c_uint8 * 2),
                ("sqrt_lambda", C.c_double)]


class IntraResult(C.Structure):
    _fields_ = [("best_mode", C.c_int32), ("pred_idx", C.c_int32), ("best_cost", C.c_double),
                ("satd", C.c_uint32 * 35), ("reserved", C.c_uint32)]


class SaoGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "rec_stride_y", "rec_stride_c", "org_stride_y", "org_stride_c")]


class SaoStats(C.Structure):
    _fields_ = [("count", C.c_int32 * 33 * 5), ("offset_org", C.c_int32 * 33 * 5)]


class MvpRequest(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("size_x", C.c_uint8), ("size_y", C.c_uint8), ("pu_mode", C.c_uint8), ("pu_idx", C.c_uint8)]


class MvpResult(C.Structure):
    _fields_ = [("amvp", (C.c_int16 * 2) * 2), ("merge", (C.c_int16 * 2) * 5), ("merge_count", C.c_int32)]


class ChromaRequest(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("size", C.c_uint8), ("luma_mode", C.c_uint8), ("reserved", C.c_uint8 * 2)]


class ChromaResult(C.Structure):
    _fields_ = [("best_idx", C.c_int32), ("best_mode", C.c_int32), ("satd_u", C.c_uint32 * 5), ("satd_v", C.c_uint32 * 5)]


class DeblockParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "stride_y", "stride_c", "tc_offset", "beta_offset", "cb_qp_offset", "cr_qp_offset")]


class InterPredRequest(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("size", C.c_int32), ("mv", C.c_int16 * 2),
                ("out_y", C.c_uint32), ("out_c", C.c_uint32)]


class IntraPredRequest(C.Structure):
    _fields_ = [("size", C.c_uint8), ("mode", C.c_uint8), ("is_luma", C.c_uint8), ("reserved", C.c_uint8), ("out", C.c_uint32)]


INTRA_REF_PITCH = 132
_lib = None


def load():
    """Load the native library. Raises XmiError when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XmiError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C x265_amd/csrc` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i = C.c_void_p, C.c_int
    L.xmi_init.restype = i
    L.xmi_init.argtypes = [i]
    L.xmi_shutdown.restype = None
    L.xmi_last_error.restype = C.c_char_p
    L.xmi_device_count.restype = i
    L.xmi_version.restype = C.c_char_p
    L.xmi_sad_surface_elems.restype = C.c_size_t
    L.xmi_sad_surface_elems.argtypes = [C.POINTER(SadGeom), i]
    L.xmi_sad_surface_index.restype = C.c_size_t
    L.xmi_sad_surface_index.argtypes = [C.POINTER(SadGeom), i, i, i, i, i]
    L.xmi_sad_surface_dev.restype = i
    L.xmi_sad_surface_dev.argtypes = [C.POINTER(SadGeom), vp, vp, vp, vp, vp, vp, i, i, vp]
    L.xmi_sad_surface.restype = i
    L.xmi_sad_surface.argtypes = [C.POINTER(SadGeom), vp, vp, i, vp, vp, vp, vp]
    if hasattr(L, "xmi_interp_planes_dev"):
        L.xmi_interp_planes_dev.restype = i
        L.xmi_interp_planes_dev.argtypes = [vp, i, i, i, vp, vp]
    if hasattr(L, "xmi_frame_create"):
        L.xmi_frame_create.restype = i
        L.xmi_frame_create.argtypes = [i, i, i, C.POINTER(vp)]
        L.xmi_frame_destroy.restype = i
        L.xmi_frame_destroy.argtypes = [vp]
        L.xmi_frame_begin.restype = i
        L.xmi_frame_begin.argtypes = [vp, vp, i, vp, i]
        L.xmi_frame_sad_surface.restype = i
        L.xmi_frame_sad_surface.argtypes = [vp, i]
        L.xmi_frame_planes_host.restype = i
        L.xmi_frame_planes_host.argtypes = [vp, vp]
        L.xmi_frame_me_search.restype = i
        L.xmi_frame_me_search.argtypes = [vp, vp, i, i, vp]
        L.xmi_frame_intra_search.restype = i
        L.xmi_frame_intra_search.argtypes = [vp, vp, vp, i, vp]
        L.xmi_frame_tu_chain.restype = i
        L.xmi_frame_tu_chain.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp]
    if hasattr(L, "xmi_intra_gather_dev"):
        L.xmi_intra_gather_dev.restype = i
        L.xmi_intra_gather_dev.argtypes = [vp, i, i, i, vp, i, vp, i, vp, vp]
    if hasattr(L, "xmi_intra_search_batch_dev"):
        L.xmi_intra_search_batch_dev.restype = i
        L.xmi_intra_search_batch_dev.argtypes = [vp, i, vp, vp, i, vp, vp]
        L.xmi_intra_search.restype = i
        L.xmi_intra_search.argtypes = [vp, C.c_uint32, vp, C.POINTER(IntraRequest), C.POINTER(IntraResult)]
    if hasattr(L, "xmi_frame_padding_dev"):
        L.xmi_frame_padding_dev.restype = i
        L.xmi_frame_padding_dev.argtypes = [vp, i, i, i, i, vp]
        L.xmi_frame_padding.restype = i
        L.xmi_frame_padding.argtypes = [vp, i, i, i, i]
    if hasattr(L, "xmi_sao_stats_dev"):
        L.xmi_sao_stats_dev.restype = i
        L.xmi_sao_stats_dev.argtypes = [C.POINTER(SaoGeom), vp, vp, vp, vp, vp, vp, vp, vp]
        L.xmi_sao_apply_plane_dev.restype = i
        L.xmi_sao_apply_plane_dev.argtypes = [vp, i, vp, i, i, i, i, vp, vp]
        L.xmi_sao_apply_plane.restype = i
        L.xmi_sao_apply_plane.argtypes = [vp, i, i, i, i, vp]
        L.xmi_sao_stats_frame.restype = i
        L.xmi_sao_stats_frame.argtypes = [C.POINTER(SaoGeom), vp, vp, vp, vp, vp, vp, vp]
    if hasattr(L, "xmi_deblock_frame_dev"):
        L.xmi_deblock_frame_dev.restype = i
        L.xmi_deblock_frame_dev.argtypes = [vp, vp, vp, C.POINTER(DeblockParams), vp, vp, vp, vp]
        L.xmi_deblock_frame.restype = i
        L.xmi_deblock_frame.argtypes = [vp, vp, vp, C.POINTER(DeblockParams), vp, vp, vp]
    if hasattr(L, "xmi_mvp_candidates_dev"):
        L.xmi_mvp_candidates_dev.restype = i
        L.xmi_mvp_candidates_dev.argtypes = [i, i, vp, vp, vp, i, vp, vp]
        L.xmi_me_requests_from_candidates_dev.restype = i
        L.xmi_me_requests_from_candidates_dev.argtypes = [i, i, vp, vp, vp, i, C.c_uint32, vp, vp]
        L.xmi_mv_field_store_dev.restype = i
        L.xmi_mv_field_store_dev.argtypes = [i, i, vp, vp, i, vp, vp, vp]
        L.xmi_fill_mvp_cand.restype = i
        L.xmi_fill_mvp_cand.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.xmi_cu_mv_clip.restype = i
        L.xmi_cu_mv_clip.argtypes = [i, i, i, i, vp, vp]
    if hasattr(L, "xmi_intra_chroma_search_batch_dev"):
        L.xmi_intra_chroma_search_batch_dev.restype = i
        L.xmi_intra_chroma_search_batch_dev.argtypes = [vp, vp, i, vp, vp, i, vp, vp]
        L.xmi_intra_chroma_search.restype = i
        L.xmi_intra_chroma_search.argtypes = [vp, vp, C.c_uint32, vp, vp, i, i, vp]
    if hasattr(L, "xmi_deblock_strengths_dev"):
        L.xmi_deblock_strengths_dev.restype = i
        L.xmi_deblock_strengths_dev.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp]
        L.xmi_deblock_strengths.restype = i
        L.xmi_deblock_strengths.argtypes = [i, i, i, vp, vp, vp, vp, vp]
    if hasattr(L, "xmi_deblock_strengths_static"):
        L.xmi_deblock_strengths_static.restype = i
        L.xmi_deblock_strengths_static.argtypes = [i, i, i, vp, vp, vp, vp, vp]
    if hasattr(L, "xmi_pred_inter_batch_dev"):
        L.xmi_pred_inter_batch_dev.restype = i
        L.xmi_pred_inter_batch_dev.argtypes = [vp, i, vp, vp, i, vp, i, vp, vp, vp, vp]
        L.xmi_intra_pred_batch_dev.restype = i
        L.xmi_intra_pred_batch_dev.argtypes = [vp, vp, i, vp, vp]
        L.xmi_pred_inter_luma.restype = i
        L.xmi_pred_inter_luma.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, i, i]
        L.xmi_pred_inter_chroma.restype = i
        L.xmi_pred_inter_chroma.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, i, i]
        L.xmi_enc_intra_pred.restype = i
        L.xmi_enc_intra_pred.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    if hasattr(L, "xmi_me_search_batch_dev"):
        L.xmi_me_search_batch_dev.restype = i
        L.xmi_me_search_batch_dev.argtypes = [vp, i, vp, i, i, i, vp, i, i, vp, vp]
        L.xmi_me_search_surface_batch_dev.restype = i
        L.xmi_me_search_surface_batch_dev.argtypes = [vp, i, vp, i, i, i, C.POINTER(SadGeom), vp, vp, vp, vp, vp, i, i, vp, vp]
    if hasattr(L, "xmi_block_metric_batch_dev"):
        L.xmi_block_metric_batch_dev.restype = i
        L.xmi_block_metric_batch_dev.argtypes = [i, i, i, vp, vp, vp, vp]
        L.xmi_tu_chain_batch_dev.restype = i
        L.xmi_tu_chain_batch_dev.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, vp]
        L.xmi_sub_dct_batch_dev.restype = i
        L.xmi_sub_dct_batch_dev.argtypes = [i, i, vp, vp, vp, vp, vp]
        L.xmi_quant_batch_dev.restype = i
        L.xmi_quant_batch_dev.argtypes = [i, i, vp, vp, vp, vp, vp, vp]
        L.xmi_dequant_batch_dev.restype = i
        L.xmi_dequant_batch_dev.argtypes = [i, i, vp, vp, vp, vp]
        L.xmi_idct_add_batch_dev.restype = i
        L.xmi_idct_add_batch_dev.argtypes = [i, i, vp, vp, vp, vp, vp]
        u32, ip = C.c_uint32, C.c_ssize_t
        L.xmi_sad.restype = C.c_int64
        L.xmi_sad.argtypes = [i, vp, u32, vp, u32]
        L.xmi_satd_hm.restype = C.c_int64
        L.xmi_satd_hm.argtypes = [i, vp, u32, vp, u32]
        L.xmi_ssd.restype = C.c_int64
        L.xmi_ssd.argtypes = [i, vp, ip, vp, ip]
        L.xmi_satd.restype = C.c_int64
        L.xmi_satd.argtypes = [i, vp, ip, vp, ip]
        L.xmi_sub_dct.restype = i
        L.xmi_sub_dct.argtypes = [vp, vp, u32, vp, u32, i, u32]
        L.xmi_quant.restype = C.c_int64
        L.xmi_quant.argtypes = [vp, vp, u32, u32, i, i, i]
        L.xmi_dequant.restype = i
        L.xmi_dequant.argtypes = [vp, vp, u32, u32, i, i]
        L.xmi_idct_add.restype = i
        L.xmi_idct_add.argtypes = [vp, u32, vp, u32, vp, i, u32]
        L.xmi_tu_chain.restype = C.c_int64
        L.xmi_tu_chain.argtypes = [vp, vp, vp, vp, i, i, i, u32, u32, i]
    _lib = L
    return L


def lib():
    return load()


def check(rc):
    """Raise XmiError with the library's message when an entry point reports failure."""
    if rc < 0:
        raise XmiError(f"libx265_amd error {rc}: {load().xmi_last_error().decode()}")
    return rc
