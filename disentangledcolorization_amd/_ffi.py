"""ctypes binding of libdisco_hip.so (include/disco_hip.h).

The product path has no CPU fallback: if the library is missing, `lib()` raises with the
command that builds it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DISCO_HIP_LIB points at another build of the same ABI (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("DISCO_HIP_LIB") or os.path.join(_HERE, "libdisco_hip.so")

OK = 0
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PREC_F16X3, PREC_MX8, PREC_MX8_ALL, PREC_X2Q, PREC_MX6 = 0, 2, 3, 4, 5
PLANE_LO, PLANE_Q, PLANE_QL, PLANE_Q6 = 1, 2, 4, 8


class DiscoError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("sp_size", C.c_int32), ("n_clusters", C.c_int32), ("random_hint", C.c_int32),
                ("precision", C.c_int32), ("network", C.c_int32), ("hint2regress", C.c_int32),
                ("spix_pos", C.c_int32), ("use_mask", C.c_int32)]


class ForwardArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("sampled_T", C.c_int32),
        ("test_mode", C.c_int32),
        ("d_gray", C.c_void_p), ("d_ab", C.c_void_p),
        ("h_init_idx", C.c_void_p), ("h_fallback_rows", C.c_void_p), ("max_fallback", C.c_int32),
        ("h_hint_pos", C.c_void_p),
        ("d_pal_logit", C.c_void_p), ("d_ref_logit", C.c_void_p), ("d_pred_colors", C.c_void_p),
        ("d_affinity", C.c_void_p), ("d_spix_colors", C.c_void_p), ("d_hint_mask", C.c_void_p),
        ("h_kmeans_events", C.c_void_p),
        ("d_workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
    ]


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("c_in0", C.c_int32),
                ("c_in1", C.c_int32), ("up0", C.c_int32), ("up1", C.c_int32), ("c_out", C.c_int32),
                ("stride", C.c_int32), ("act", C.c_int32), ("slope", C.c_float), ("precision", C.c_int32),
                ("sexp_in", C.c_int32), ("sexp_out", C.c_int32), ("sexp_res", C.c_int32)]


class ConvMxDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("c_in0", C.c_int32), ("c_in1", C.c_int32),
                ("up0", C.c_int32), ("up1", C.c_int32), ("sexp0", C.c_int32), ("sexp1", C.c_int32), ("c_out", C.c_int32),
                ("stride", C.c_int32), ("act", C.c_int32), ("slope", C.c_float), ("out_planes", C.c_int32),
                ("out_sexp", C.c_int32), ("out_f32", C.c_int32), ("res_planes", C.c_int32), ("res_sexp", C.c_int32), ("x2q", C.c_int32),
                ("q6", C.c_int32), ("d2s", C.c_int32)]


# name -> (restype, argtypes); every symbol include/disco_hip.h declares
_P, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
SIGNATURES = {
    "disco_abi_version": (_I, []),
    "disco_last_error": (C.c_char_p, []),
    "disco_create": (_I, [_I, C.POINTER(Options), C.POINTER(_P)]),
    "disco_destroy": (_I, [_P]),
    "disco_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "disco_finalize": (_I, [_P]),
    "disco_expected_tensors": (_I, []),
    "disco_expected_tensor_ctx": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "disco_expected_tensor": (_I, [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(_I)]),
    "disco_workspace_bytes": (_I, [_P, _I, _I, _I, _I, C.POINTER(_SZ)]),
    "disco_forward": (_I, [_P, C.POINTER(ForwardArgs)]),
    "disco_calibrate": (_I, [_P, _P, _I, _I, _I]),
    "disco_saturation_count": (_I, [_P, _P, C.POINTER(C.c_uint64)]),
    "disco_kmeans_fallback_count": (_I, [_P, _P, C.POINTER(C.c_uint64)]),
    "disco_op_kmeans_fallbacks": (_I, [_P, _I, _I, _P, C.POINTER(C.c_int)]),
    "disco_calibration_count": (_I, [_P]),
    "disco_enhance_arithmetic": (_I, [_P, C.POINTER(_I), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "disco_calibration_entry": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(_I)]),
    "disco_forward_segnet": (_I, [_P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "disco_forward_repnet": (_I, [_P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "disco_forward_enhance": (_I, [_P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "disco_subnet_workspace_bytes": (_I, [_P, _I, _I, _I, _I, C.POINTER(_SZ)]),
    "disco_sync": (_I, [_P]),
    "disco_set_profiling": (_I, [_P, _I]),
    "disco_set_progress_event": (_I, [_P, _P, _I]),
    "disco_set_debug_checksums": (_I, [_P, _P, _I, _I]),
    "disco_set_debug_dump": (_I, [_P, _P, _SZ]),
    "disco_profile_count": (_I, [_P]),
    "disco_profile_entry": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "disco_profile_conv": (_I, [_P, C.POINTER(_I), C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "disco_profile_conv_bytes": (_I, [_P, C.POINTER(C.c_double)]),
    "disco_profile_conv_entry": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "disco_op_nchw_to_act": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "disco_op_act_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "disco_op_conv3x3_pack": (_I, [_P, _I, _I, _P, C.POINTER(_SZ)]),
    "disco_op_conv3x3": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "disco_op_act_bytes": (_I, [_I, _I, _I, _I, _I, C.POINTER(_SZ)]),
    "disco_op_gray_tail": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "disco_op_nchw_to_act_mx": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "disco_op_act_mx_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "disco_op_conv3x3_mx_pack": (_I, [_P, _I, _I, _I, _P, _P, C.POINTER(_SZ)]),
    "disco_op_conv3x3_mx": (_I, [C.POINTER(ConvMxDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "disco_op_conv3x3_tapmask": (_I, [_P, _I, _I, _P]),
    "disco_diag_mfma_rate": (_I, [_I, _I, C.POINTER(C.c_double)]),
    "disco_op_deconv4x4_pack": (_I, [_P, _I, _I, _P, C.POINTER(_SZ)]),
    "disco_op_deconv4x4": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float, _I, _P]),
    "disco_op_poolfeat": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "disco_op_poolfeat_act": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P]),
    "disco_op_upfeat": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "disco_op_encoder_weight_floats": (_SZ, []),
    "disco_op_encoder_stack": (_I, [_P, _P, _P, _P, _I, _I, _P, _SZ, _P]),
    "disco_op_encoder_stack_masked": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _SZ, _P]),
    "disco_op_kmeans_anchors": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "disco_op_kmeans_workspace_bytes": (_SZ, [_I, _I]),
    "disco_op_kmeans_anchors_ws": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "disco_op_select_colors": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "disco_op_nearest_bin": (_I, [_P, _P, _I, _I, _P]),
    "disco_op_position_encoding": (_I, [_P, _I, _I, _P]),
    "disco_op_decode_ind2ab": (_I, [_P, _P, _I, _I, _I, _P]),
    "disco_op_decode_annealed": (_I, [_P, _P, _I, _I, C.c_float, _P]),
    "disco_op_rgb2lab": (_I, [_P, _P, _I, _I, _I, _P]),
    "disco_op_lab2rgb": (_I, [_P, _P, _I, _I, _I, _P]),
    "disco_op_rgb8_to_lab": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "disco_op_lab_to_rgb8": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "disco_op_rgb8_resize_to_lab": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "disco_op_mark_color_hints": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
}

ABI_VERSION = 11      # include/disco_hip.h DISCO_ABI_VERSION this binding matches (struct layouts, entry points)
_lib = None


def lib():
    """The loaded library (loads once).  Raises DiscoError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DiscoError(
                "libdisco_hip.so is missing (%s). The HIP extension is required — there is no CPU "
                "fallback. Build it with: python -m disentangledcolorization_amd.build" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = handle.disco_abi_version()
        if got != ABI_VERSION:          # a stale build next to newer Python sources (struct layouts differ between versions): refuse it
            raise DiscoError("libdisco_hip.so reports ABI version %d, this binding was written for %d: rebuild it "
                             "(python -m disentangledcolorization_amd.build --force)" % (got, ABI_VERSION))
        _lib = handle
    return _lib


def check(rc):
    if rc != OK:
        raise DiscoError("libdisco_hip error %d: %s" % (rc, lib().disco_last_error().decode()))


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array, or None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
