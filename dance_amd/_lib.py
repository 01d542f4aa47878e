"""ctypes binding of libdance_hip.so (C ABI: include/dance_hip.h).

There is NO fallback: if the shared object is missing or a symbol is absent, importing
`lib()` raises.  The product path never routes through torch ops or the oracle.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DANCE_HIP_LIB: load another build of the same library (A/B kernel experiments)
SO_PATH = os.environ.get("DANCE_HIP_LIB") or os.path.join(_HERE, "libdance_hip.so")

ABI_VERSION = 16
EVOLVE_EVENTS = 14          # DANCE_EVOLVE_EVENTS
STAGE_NAMES = ("head", "res0", "res1", "res2", "res3", "res4", "res5", "res6", "fusion_max",
               "row_constant", "pred0", "pred1", "offset_update")


class SnakeNetStruct(ctypes.Structure):
    """Mirror of `dance_snake_net` (include/dance_hip.h)."""

    _fields_ = [
        ("head_w", c_void_p), ("head_b", c_void_p), ("head_scale", c_void_p), ("head_shift", c_void_p),
        ("res_w", c_void_p * 7), ("res_b", c_void_p * 7), ("res_scale", c_void_p * 7),
        ("res_shift", c_void_p * 7),
        ("fusion_w", c_void_p), ("fusion_b", c_void_p),
        ("pred0_wg", c_void_p),
        ("pred0_ws", c_void_p), ("pred0_b", c_void_p),
        ("pred1_w", c_void_p), ("pred1_b", c_void_p),
        ("pred2_w", c_void_p), ("pred2_b", c_void_p),
        ("dilation", c_int * 7),
        ("feat_channels", c_int),
    ]


class SnakeNet16Struct(ctypes.Structure):
    """Mirror of `dance_snake_net16` (include/dance_hip.h)."""

    _fields_ = [
        ("head_w", c_void_p), ("head_bs", c_void_p), ("head_scale", c_void_p), ("head_shift", c_void_p),
        ("res_w", c_void_p * 7), ("res_bs", c_void_p * 7), ("res_scale", c_void_p * 7),
        ("res_shift", c_void_p * 7),
        ("fuse_w", c_void_p), ("fuse_bs", c_void_p), ("pred0_wg", c_void_p), ("pred0_b", c_void_p),
        ("pred1_w", c_void_p), ("pred1_bs", c_void_p),
        ("pred2_w", c_void_p), ("pred2_b", c_void_p),
        ("head_winv", c_float), ("res_winv", c_float * 7), ("res_rmul", c_float * 7), ("fuse_winv", c_float),
        ("pred0_winv", c_float), ("pred1_winv", c_float),
        ("dilation", c_int * 7),
        ("feat_channels", c_int),
        ("head_wd", c_void_p),
        ("head_wd_winv", c_float),
    ]


class PolicyStruct(ctypes.Structure):
    """Mirror of `dance_policy` (include/dance_hip.h); the defaults are the library's."""

    _fields_ = [("few_contour_forms", c_int), ("fuse_tail_split", c_int), ("tail_group", c_int),
                ("coop_stack", c_int)]

    def __init__(self, few_contour_forms=1, fuse_tail_split=1, tail_group=0, coop_stack=1):
        super().__init__(int(few_contour_forms), int(fuse_tail_split), int(tail_group), int(coop_stack))


ABSMAX_WORDS = 8            # DANCE_ABSMAX_WORDS
ABSMAX_OF_LAYOUT, ABSMAX_OF_GN_RELU, ABSMAX_OF_FEATURES = 0, 1, 2


class InitFuse16Struct(ctypes.Structure):
    """Mirror of `dance_init_fuse16` (include/dance_hip.h)."""

    _fields_ = [("w16", c_void_p), ("wg_center", c_void_p), ("bias", c_void_p), ("w_inv_scale", c_float)]


class FcoseLevelStruct(ctypes.Structure):
    """Mirror of `dance_fcose_level` (include/dance_hip.h)."""

    _fields_ = [("cls", c_void_p), ("ctr", c_void_p), ("reg", c_void_p), ("ext", c_void_p),
                ("locations", c_void_p), ("hw", c_int), ("stride", c_float)]


P = c_void_p
# name -> (restype, argtypes); every symbol include/dance_hip.h declares
SIGNATURES = {
    "dance_abi_version": (c_int, []),
    "dance_last_error": (c_char_p, []),
    "dance_device_info": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "dance_collect_extreme_point": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dance_calculate_edge_num": (c_int, [P, P, P, c_int, c_int, c_int, c_int64, P]),
    "dance_calculate_wnp": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dance_roll_array": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "dance_nchw_to_p16img": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, P]),
    "dance_nhwc_to_p16img": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, P]),
    "dance_upsample2x_add": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "dance_conv3x3_16": (c_int, [P, P, c_float, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "dance_conv1x1_nhwc": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "dance_conv3x3_16_partial_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "dance_gn_relu": (c_int, [P, P, P, P, c_float, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dance_nms_sorted": (c_int, [P, c_int, c_float, P, P, P, P]),
    "dance_nms_sorted_plain": (c_int, [P, c_int, c_float, P, P, P, P]),
    "dance_nms_sorted_plain_static": (c_int, [P, c_int, c_float, P, P, P]),
    "dance_nms_segments_plain": (c_int, [P, P, c_int, c_int, c_float, P, P, P, P]),
    "dance_contour_upsample_2x": (c_int, [P, P, c_int, c_int, P]),
    "dance_attender_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "dance_attender": (c_int, [P, P, P, P, P, P, P, P, P, c_float, P, P, c_int64, c_int, c_int, c_int, P]),
    "dance_polygon_rle": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P]),
    "dance_rle_to_string": (c_int, [P, P, c_int, c_int, P, P, c_int64, P, P]),
    "dance_rle_decode": (c_int, [P, P, c_int, c_int, P, c_int, c_int, P]),
    "dance_fcose_count": (c_int, [P, c_int, P, c_int, c_int, c_float, c_int, P]),
    "dance_fcose_fill": (c_int, [P, c_int, P, P, c_int, c_int, c_float, c_int, P, P, P, P, P, P, P]),
    "dance_uniform_sample": (c_int, [P, P, P, P, P, P, c_int, P]),
    "dance_dense_targets": (c_int, [P, P, P, c_int, P, P, c_int, P]),
    "dance_uniform_sample_1d": (c_int, [P, P, P, P, P, P, c_int, P]),
    "dance_aux_extreme_points": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "dance_segment_matching": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dance_multisegment_matching": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dance_uniform_upsample": (c_int, [c_int, P, c_int, c_int, c_int, P, P, P, P]),
    "dance_octagon_upsample": (c_int, [P, c_int64, c_int64, P, P, c_int, c_float, c_float, c_int, c_int, P, P, P, P, P]),
    "dance_nchw_to_nhwc": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dance_feature_absmax": (c_int, [P, c_int, c_int64, P, P, P]),
    "dance_absmax_partials": (c_int64, [c_int, c_int, c_int64]),
    "dance_gather_features": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_int, P]),
    "dance_circconv": (c_int, [P, c_int64, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, P]),
    "dance_gather_circconv_head": (c_int, [P, P, P, P, P, P, P, P, P, c_int64, P, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "dance_row_constant": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "dance_offset_update": (c_int, [P, c_int64, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float,
                                    c_float, P]),
    "dance_snake_workspace_bytes": (c_int64, [c_int, c_int]),
    "dance_snake_evolve": (c_int, [POINTER(SnakeNetStruct), P, P, P, P, P, P, P, P, c_int64, c_int,
                                   c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_float, c_float, c_int, POINTER(c_void_p), P]),
    "dance_conv16": (c_int, [P, c_int64, c_int64, P, P, c_float, P, P, P, c_int, c_int, c_int, P,
                             c_int64, c_int64, P, c_int64, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "dance_fuse16_parts": (c_int, [c_int, c_int, POINTER(PolicyStruct)]),
    "dance_fuse16w_supported": (c_int, [c_int, c_int]),
    "dance_fuse16w_tail_split": (c_int, [c_int, c_int, POINTER(PolicyStruct)]),
    "dance_fuse16w_raw_split": (c_int, [P, c_int64, c_int64, P, P, c_float, c_float, P, c_int64, c_int64, P, c_int,
                                        c_int, c_int, c_int, P]),
    "dance_fuse16w_raw": (c_int, [P, c_int64, c_int64, P, P, c_float, c_float, P, c_int64, c_int64, P, c_int, c_int,
                                  c_int, c_int, P]),
    "dance_tail16": (c_int, [P, c_int64, c_int64, P, c_int, P, P, P, P, c_float, P, P, P, P, P, P, c_int,
                             c_int, c_int, c_float, c_float, c_int, POINTER(PolicyStruct), P]),
    "dance_gather_head16w": (c_int, [P, P, P, P, P, P, P, c_float, c_float, P, P, P, c_int64, P, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, POINTER(PolicyStruct), P]),
    "dance_res_stack16": (c_int, [POINTER(SnakeNet16Struct), P, c_int64, c_int64, c_int, c_int, c_int,
                                  POINTER(PolicyStruct), P, P]),
    "dance_init_head16": (c_int, [P, P, P, P, c_float, P, P, P, P, P, c_float, c_float, P, P, P, c_int64, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "dance_snake_init16": (c_int, [POINTER(SnakeNet16Struct), POINTER(InitFuse16Struct), P, P, P, P, P,
                                   c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P,
                                   POINTER(PolicyStruct), POINTER(c_void_p), P]),
    "dance_snake_evolve16": (c_int, [POINTER(SnakeNet16Struct), P, P, P, P, P, P, P, P, c_int64, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                     c_int, POINTER(PolicyStruct), POINTER(c_void_p), P]),
}

_lib = None


def lib():
    """The loaded library; raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "dance_amd: %s is missing -- build it with `python -m dance_amd.build` "
            "(there is no CPU or torch fallback for the contour path)" % SO_PATH)
    h = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = h.dance_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("dance_amd: libdance_hip.so ABI %d, bindings expect %d" % (got, ABI_VERSION))
    _lib = h
    return h


def check(code, what):
    """Turn a non-zero return code into RuntimeError (the reference's AT_ASSERTM /
    THCudaCheck failures surface as RuntimeError too, utils.cu:162-209)."""
    if code != 0:
        msg = lib().dance_last_error()
        raise RuntimeError("%s failed (hipError %d): %s" % (what, code, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream
