"""ctypes loader for kvazaar_amd/libkvzhip.so (the C ABI of include/kvz_hip.h).

There is no fallback: if the library is missing or no MI355X is usable, the
calls raise.  Nothing in this package imports anything from oracle/."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkvzhip.so")
_LIB = None

OK = 0


class BlockPair(C.Structure):
    """kvz_hip_block_pair"""
    _fields_ = [("x1", C.c_int32), ("y1", C.c_int32), ("x2", C.c_int32), ("y2", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32)]


class IpolBlock(C.Structure):
    """kvz_hip_ipol_block"""
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("mv_frac_x", C.c_int32), ("mv_frac_y", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32)]


class QuantParams(C.Structure):
    """kvz_hip_quant_params"""
    _fields_ = [("qp", C.c_int32), ("slice_is_intra", C.c_int32), ("signhide", C.c_int32),
                ("scaling_list", C.c_int32), ("quant_coeff", C.c_void_p), ("dequant_coeff", C.c_void_p)]


class KvzHipError(RuntimeError):
    pass


_P = C.c_void_p
_SZ = C.c_size_t
_I = C.c_int
_U = C.c_uint32

# name -> (restype, argtypes); every symbol include/kvz_hip.h declares
SIGNATURES = {
    "kvz_hip_init": (_I, [_I]),
    "kvz_hip_set_device": (_I, [_I]),
    "kvz_hip_get_device": (_I, []),
    "kvz_hip_shutdown": (None, []),
    "kvz_hip_device_count": (_I, []),
    "kvz_hip_last_error": (C.c_char_p, []),
    "kvz_hip_device_name": (C.c_char_p, []),
    "kvz_hip_abi_version": (_I, []),
    "kvz_hip_set_tuning": (_I, [C.c_char_p, _I]),
    "kvz_hip_malloc": (_P, [_SZ]),
    "kvz_hip_free": (None, [_P]),
    "kvz_hip_malloc_host": (_P, [_SZ]),
    "kvz_hip_free_host": (None, [_P]),
    "kvz_hip_memcpy_h2d": (_I, [_P, _P, _SZ, _P]),
    "kvz_hip_memcpy_d2h": (_I, [_P, _P, _SZ, _P]),
    "kvz_hip_memset": (_I, [_P, _I, _SZ, _P]),
    "kvz_hip_memcpy_d2d": (_I, [_P, _P, _SZ, _P]),
    "kvz_hip_memcpy_peer": (_I, [_P, _I, _P, _I, _SZ, _P]),
    "kvz_hip_halo_exchange": (_I, [_P, _P, _P, _U, _I, _P]),
    "kvz_hip_stream_create": (_P, []),
    "kvz_hip_stream_destroy": (None, [_P]),
    "kvz_hip_stream_sync": (_I, [_P]),
    "kvz_hip_event_create": (_P, []),
    "kvz_hip_event_destroy": (None, [_P]),
    "kvz_hip_event_record": (_I, [_P, _P]),
    "kvz_hip_event_elapsed_ms": (_I, [_P, _P, C.POINTER(C.c_float)]),
    "kvz_hip_stream_wait_event": (_I, [_P, _P]),
    "kvz_hip_graph_begin": (_I, [_P]),
    "kvz_hip_graph_end": (_I, [_P, C.POINTER(_P)]),
    "kvz_hip_graph_launch": (_I, [_P, _P]),
    "kvz_hip_graph_destroy": (None, [_P]),
    "kvz_hip_sad_nxn_batch": (_I, [_I, _P, _P, _SZ, _P, _P]),
    "kvz_hip_satd_nxn_batch": (_I, [_I, _P, _P, _SZ, _P, _P]),
    "kvz_hip_sad_nxn_dual_batch": (_I, [_I, _P, _SZ, _SZ, _P, _SZ, _P, _P]),
    "kvz_hip_satd_nxn_dual_batch": (_I, [_I, _P, _SZ, _SZ, _P, _SZ, _P, _P]),
    "kvz_hip_reg_sad_batch": (_I, [_P, _U, _P, _U, _P, _SZ, _P, _P]),
    "kvz_hip_image_calc_sad_batch": (_I, [_P, _U, _P, _U, _I, _I, _P, _SZ, _P, _P]),
    "kvz_hip_image_calc_satd_batch": (_I, [_P, _U, _P, _U, _I, _I, _P, _SZ, _P, _P]),
    "kvz_hip_pixels_calc_ssd_batch": (_I, [_P, _U, _P, _U, _P, _SZ, _P, _P]),
    "kvz_hip_satd_any_size_quad_batch": (_I, [_P, _U, _SZ, _P, _U, _P, _SZ, _P, _P]),
    "kvz_hip_ctu_sad_grid_batch": (_I, [_P, _U, _I, _I, _P, _U, _I, _I, _P, _SZ, _P, _I, _P, _P]),
    "kvz_hip_bipred_blend_batch": (_I, [_I, _I, _I, _P, _I, _P, _P, _SZ, _P]),
    "kvz_hip_transform_batch": (_I, [_I, _I, _P, _P, _SZ, _P]),
    "kvz_hip_quant_batch": (_I, [C.POINTER(QuantParams), _P, _P, _I, _I, _I, _SZ, _P]),
    "kvz_hip_dequant_batch": (_I, [C.POINTER(QuantParams), _P, _P, _I, _I, _SZ, _P]),
    "kvz_hip_coeff_abs_sum_batch": (_I, [_P, _SZ, _SZ, _P, _P]),
    "kvz_hip_quantize_residual_batch": (_I, [C.POINTER(QuantParams), _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "kvz_hip_residual_batch": (_I, [_P, _P, _P, _SZ, _P]),
    "kvz_hip_reconstruct_batch": (_I, [_P, _P, _P, _SZ, _P]),
    "kvz_hip_quantize_residual_cost_batch": (_I, [C.POINTER(QuantParams), _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "kvz_hip_sample_luma_batch": (_I, [_P, _U, _I, _I, _P, _P, _SZ, _I, _P, _P]),
    "kvz_hip_sample_chroma_batch": (_I, [_P, _U, _I, _I, _P, _P, _SZ, _I, _P, _P]),
    "kvz_hip_search_frac_batch": (_I, [_P, _U, _P, _U, _I, _I, _P, _SZ, _P, _P, _P]),
    "kvz_hip_search_pu_batch": (_I, [_P, _U, _I, _I, _P, _U, _I, _I, _P, _SZ, _P, _P, _P]),
    "kvz_hip_search_pu_multi_batch": (_I, [_P, _U, _I, _I, _P, _U, _I, _I, _I, _P, _SZ, _P, _P, _P]),
    "kvz_hip_me_service_create": (_P, [_P]),
    "kvz_hip_me_service_destroy": (None, [_P]),
    "kvz_hip_me_service_put_rect": (_I, [_P, _I, _P, _U, _I, _I, _I, _I]),
    "kvz_hip_me_service_search": (_I, [_P, _P, _P]),
    "kvz_hip_me_service_get_stats": (_I, [_P, _P]),
    "kvz_hip_me_service_plane": (_P, [_P, _I]),
    "kvz_hip_me_service_sad_tables": (_P, [_P, _I, _I, _P, _I, _I, _I]),
    "kvz_hip_bipred_cost_batch": (_I, [_P, _U, _I, _I, _P, _U, _P, _U, _I, _I, _P, _SZ, _P, _P]),
    "kvz_hip_inter_candidates_batch": (_I, [_P, _P, _P, _P, _P, _SZ, _P, _P]),
    "kvz_hip_inter_candidates_multi_batch": (_I, [_P, _I, _P, _SZ, _P, _P]),
    "kvz_hip_intra_build_reference_batch": (_I, [_I, _I, _P, _I, _I, _I, _P, _SZ, _P, _P]),
    "kvz_hip_intra_predict_batch": (_I, [_I, _I, _P, _SZ, _P, _I, _P, _P]),
    "kvz_hip_intra_rough_batch": (_I, [_I, _I, _P, _P, _SZ, _P, _P, _P]),
    "kvz_hip_sao_edge_stats_batch": (_I, [_P, _P, _I, _I, _SZ, _P, _P]),
    "kvz_hip_sao_edge_ddistortion_batch": (_I, [_P, _P, _I, _I, _SZ, _P, _P, _P]),
    "kvz_hip_sao_band_stats_batch": (_I, [_P, _P, _I, _I, _SZ, _P, _P]),
    "kvz_hip_sao_band_ddistortion_batch": (_I, [_P, _P, _I, _I, _SZ, _P, _P, _P, _P]),
    "kvz_hip_sao_reconstruct_color_batch": (_I, [_P, _U, _I, _I, _P, _U, _P, _SZ, _P, _I, _I, _P]),
    "kvz_hip_deblock_frame": (_I, [_P, _U, _P, _P, _U, _I, _I, _P, _P, _P]),
    "kvz_hip_set_registrar": (None, [_P]),
    "kvz_hip_dropin_calls": (C.c_ulonglong, []),
    "kvz_hip_set_state_accessors": (None, [_P]),
    "kvz_strategy_register_picture_hip": (_I, [_P, C.c_uint8]),
    "kvz_strategy_register_dct_hip": (_I, [_P, C.c_uint8]),
    "kvz_strategy_register_quant_hip": (_I, [_P, C.c_uint8]),
    "kvz_strategy_register_ipol_hip": (_I, [_P, C.c_uint8]),
    "kvz_strategy_register_intra_hip": (_I, [_P, C.c_uint8]),
    "kvz_strategy_register_sao_hip": (_I, [_P, C.c_uint8]),
}


def load(path=None):
    """dlopen the library and bind every exported entry point.  Does not touch the GPU."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = path or LIB_PATH
    # One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64; if this library is loaded first it brings in the
    # system's copy and a later `import torch` in the same process fails to see the device ("no ROCm-capable device").  A Python
    # host that has torch gets it loaded first, so that both share the runtime torch ships (a C host is not concerned).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise KvzHipError("%s not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                          "kvazaar_amd has no CPU fallback" % path)
    L = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)          # AttributeError if the symbol is not exported
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def check(rc, what=""):
    if rc != OK:
        L = load()
        raise KvzHipError("%s failed (rc=%d): %s" % (what or "kvz_hip call", rc, (L.kvz_hip_last_error() or b"").decode()))


def init(device=-1):
    L = load()
    check(L.kvz_hip_init(device), "kvz_hip_init")
    return L
