"""ctypes binding of libcapsyolo_hip.so (C-ABI declared in include/capsyolo_hip.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CAPSYOLO_LIB') or os.path.join(_HERE, 'libcapsyolo_hip.so')   # CAPSYOLO_LIB: developer override

_c_float_p = C.c_void_p      # device pointers travel as integers
_ll = C.c_longlong


class ConvGemm(C.Structure):
    _fields_ = [('X', C.c_void_p), ('Wp', C.c_void_p), ('Y', C.c_void_p), ('bias', C.c_void_p), ('stats', C.c_void_p),
                ('xs_b', _ll), ('xs_y', _ll), ('xs_x', _ll), ('xs_c', _ll),
                ('B', C.c_int), ('Hi', C.c_int), ('Wi', C.c_int), ('Cin', C.c_int),
                ('Ho', C.c_int), ('Wo', C.c_int), ('N', C.c_int),
                ('TH', C.c_int), ('TW', C.c_int), ('in_stride', C.c_int), ('dy0', C.c_int), ('dx0', C.c_int),
                ('dstep', C.c_int),
                ('Hy', C.c_int), ('Wy', C.c_int), ('out_stride', C.c_int), ('out_oy', C.c_int), ('out_ox', C.c_int),
                ('act', C.c_int),
                ('bn_z', C.c_void_p), ('bn_scale', C.c_void_p), ('bn_shift', C.c_void_p), ('bn_mean', C.c_void_p),
                ('bn_invstd', C.c_void_p), ('bn_red', C.c_void_p), ('bn_slope', C.c_float), ('act_slope', C.c_float),
                ('ws', C.c_void_p), ('ws_floats', _ll)]


class ConvWgrad(C.Structure):
    _fields_ = [('X', C.c_void_p), ('dZ', C.c_void_p), ('dW', C.c_void_p), ('slabs', C.c_void_p),
                ('xs_b', _ll), ('xs_y', _ll), ('xs_x', _ll), ('xs_c', _ll),
                ('B', C.c_int), ('Hi', C.c_int), ('Wi', C.c_int), ('Cin', C.c_int),
                ('Ho', C.c_int), ('Wo', C.c_int), ('N', C.c_int),
                ('KH', C.c_int), ('KW', C.c_int), ('stride', C.c_int), ('pad', C.c_int)]


class RoutingFwd(C.Structure):
    _fields_ = [('u', C.c_void_p), ('W', C.c_void_p), ('v_out', C.c_void_p), ('s_hist', C.c_void_p),
                ('R', C.c_int), ('N', C.c_int), ('C', C.c_int), ('Din', C.c_int), ('Dout', C.c_int),
                ('n_iter', C.c_int), ('gather_g', C.c_int), ('gather_B', C.c_int), ('ws', C.c_void_p)]


class RoutingBwd(C.Structure):
    _fields_ = [('u', C.c_void_p), ('W', C.c_void_p), ('s_hist', C.c_void_p), ('dv', C.c_void_p),
                ('du', C.c_void_p), ('dW', C.c_void_p), ('ws', C.c_void_p),
                ('R', C.c_int), ('N', C.c_int), ('C', C.c_int), ('Din', C.c_int), ('Dout', C.c_int),
                ('n_iter', C.c_int), ('gather_g', C.c_int), ('gather_B', C.c_int)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argtypes (all return int unless listed in _RET)
_SIGS = {
    'cy_conv_pack_weights': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'cy_conv_gemm': [C.POINTER(ConvGemm), _P],
    'cy_conv_wgrad': [C.POINTER(ConvWgrad), _P],
    'cy_channel_sum': [_P, _P, _L, _I, _P],
    'cy_wino_pack_weights': [_P, _P, _I, _I, _I, _P],
    'cy_conv1_3x3_fwd': [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P],
    'cy_conv1_3x3_stats': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'cy_conv1_bn_bwd_onepass': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'cy_conv1_bn_bwd_onepass_bf16': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'cy_conv1_3x3_wgrad': [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'cy_conv1_bn_bwd_reduce': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _P],
    'cy_conv1_bn_bwd_reduce_bf16': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _P],
    'cy_conv1_bn_bwd_wgrad': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _L, _P, _P, _I, _I, _I, _I, _P],
    'cy_conv1_bn_bwd_wgrad_bf16': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _L, _P, _P, _I, _I, _I, _I, _P],
    'cy_conv1_3x3_fwd_act_bf16': [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P],
    'cy_conv3x3_winograd': [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    'cy_conv3x3_winograd_ws': [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P, _L, _P],
    'cy_wino4_pack_weights': [_P, _P, _I, _I, _I, _P],
    'cy_conv3x3_winograd4': [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    'cy_conv3x3_winograd4_wgrad': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'cy_conv3x3_winograd4_wgrad_bn': [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    'cy_conv3x3_winograd_wgrad': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'cy_conv3x3_winograd_wgrad_bn': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    'cy_bn_param_grad': [_P, _P, _P, _I, _P],
    'cy_wino2_pack_weights': [_P, _P, _I, _I, _P],
    'cy_conv4x4s2_winograd': [_P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _P],
    'cy_conv4x4s2_winograd4': [_P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _P],
    'cy_wino4s2_pack_weights': [_P, _P, _I, _I, _P],
    'cy_wino4s2_pack_dgrad_weights': [_P, _P, _I, _I, _P],
    'cy_conv4x4s2_winograd4_dgrad': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P],
    'cy_conv4x4s2_winograd_wgrad': [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    'cy_wino2_pack_dgrad_weights': [_P, _P, _I, _I, _P],
    'cy_conv4x4s2_winograd_dgrad': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P],
    'cy_bn_finalize': [_P, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P, _P],
    'cy_bn_red_fold': [_P, _I, C.c_double, _P, _P, _P, _I, _P],
    'cy_bn_eval_scale_shift': [_P, _P, _P, _P, _F, _P, _P, _I, _P],
    'cy_bn_fold_eval': [_P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P],
    'cy_affine_act': [_P, _P, _P, _P, _F, _L, _I, _P],
    'cy_bn_bwd_reduce': [_P, _P, _P, _P, _P, _P, _F, _P, _L, _I, _P],
    'cy_bn_bwd_apply': [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _L, _I, _P],
    'cy_act_bwd': [_P, _P, _P, _F, _L, _P],
    'cy_routing_fwd': [C.POINTER(RoutingFwd), _P],
    'cy_routing_bwd': [C.POINTER(RoutingBwd), _P],
    'cy_squash_fwd': [_P, _P, _L, _I, _P],
    'cy_squash_bwd': [_P, _P, _P, _L, _I, _P],
    'cy_length_fwd': [_P, _P, _L, _I, _P],
    'cy_length_bwd': [_P, _P, _P, _P, _L, _I, _P],
    'cy_darkcapsule_loss': [_P, _P, _I, _P, _P, _I, _I, _P],
    'cy_darkcapsule2_loss': [_P, _P, _P, _P, _I, _I, _I, _P],
    'cy_darkcapsule3_loss': [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'cy_margin_loss': [_P, _P, _P, _P, _I, _I, _P],
    'cy_recon_loss_add': [_P, _P, _F, _P, _P, _L, _P],
    'cy_dark_loss': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P],
    'cy_scale_by_device_scalar': [_P, _P, _P, _L, _P],
    'cy_center_u8': [_P, _P, _I, _I, _I, _I, _I, _P],
    'cy_permute4': [_P, _P, _L, _I, _I, _I, _L, _L, _L, _L, _I, _P],
    'cy_maxpool2_fwd': [_P, _P, _P, _I, _I, _I, _I, _P],
    'cy_affine_act_maxpool2': [_P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P],
    'cy_maxpool2_bwd_bn': [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P],
    'cy_maxpool2_bwd': [_P, _P, _P, _I, _I, _I, _I, _P],
    'cy_upsample_fwd': [_P, _P, _I, _I, _I, _I, _I, _P],
    'cy_upsample_bwd': [_P, _P, _I, _I, _I, _I, _I, _P],
    'cy_tanh_fwd': [_P, _P, _L, _P],
    'cy_tanh_bwd': [_P, _P, _P, _L, _P],
    'cy_yolo_head_fwd': [_P, _P, _L, _I, _I, _P],
    'cy_yolo_head_bwd': [_P, _P, _P, _L, _I, _I, _P],
    'cy_yolo_decode_boxes': [_P, _P, C.c_double, C.c_double, _I, _I, _I, _I, _F, _P, _P, _P, _P, _I, _P],
    'cy_detect_confusion': [_P, _P, _I, _P, _P, _I, _I, C.c_double, _I, _P, _P],
    'cy_pick_capsule': [_P, _P, _P, _I, _I, _I, _I, _P],
    'cy_zero_bytes': [_P, _L, _P],
    'cy_conv_bf16_pack_weights': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'cy_conv_gemm_bf16': [C.POINTER(ConvGemm), _I, _P],
    'cy_conv_gemm_bf16_classes': [C.POINTER(ConvGemm), _I, _I, _P],
    'cy_conv_wgrad_bf16': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'cy_conv_wgrad_bf16_bn': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'cy_affine_act_bf16': [_P, _P, _P, _P, _F, _L, _I, _I, _P],
    'cy_bn_bwd_reduce_bf16': [_P, _P, _I, _P, _P, _P, _P, _F, _P, _L, _I, _P],
    'cy_bn_bwd_apply_bf16': [_P, _P, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _L, _I, _P],
    'cy_cast_f32_bf16': [_P, _P, _L, _P],
    'cy_cast_bf16_f32': [_P, _P, _L, _P],
    'cy_multi_copy': [_P, _P, _I, _I, _P, _I, _F, _P],
    'cy_adam_multi': [_P, _P, _I, _I, _F, _F, _F, _F, _F, _F, _P],
    'cy_adam_multi_dev': [_P, _P, _I, _I, _P, _P],
}
_RET = {
    'capsyolo_last_error': (C.c_char_p, []),
    'capsyolo_abi_version': (C.c_int, []),
    'cy_conv_packed_floats': (_L, [_I, _I]),
    'cy_conv_bf16_packed_elems': (_L, [_I, _I]),
    'cy_conv_wgrad_bf16_ws_floats': (_L, [_I, _I, _I, _I, _I, _I, _I]),
    'cy_conv_wgrad_bf16_bn_ws_floats': (_L, [_I, _I, _I, _I, _I, _I, _I]),
    'cy_wino_packed_floats': (_L, [_I, _I]),
    'cy_wino_split_ws_floats': (_L, [_I, _I, _I, _I, _I, _I]),
    'cy_wino4_packed_floats': (_L, [_I, _I]),
    'cy_wino4_wgrad_ws_floats': (_L, [_I, _I, _I, _I, _I]),
    'cy_wino4_wgrad_ok': (_I, [_I, _I, _I, _I, _I]),
    'cy_conv1_3x3_wgrad_ws_floats': (_L, [_I, _I, _I, _I]),
    'cy_conv1_3x3_stats_ws_floats': (_L, [_I, _I]),
    'cy_conv1_3x3_stats_m2_offset': (_L, [_I, _I]),
    'cy_conv1_bn_bwd_wgrad_ws_floats': (_L, [_I, _I, _I, _I]),
    'cy_wino2_packed_floats': (_L, [_I, _I]),
    'cy_wino4s2_packed_floats': (_L, [_I, _I]),
    'cy_wino4s2_ok': (_I, [_I, _I, _I, _I, _I]),
    'cy_wino4s2_dgrad_ok': (_I, [_I, _I, _I, _I, _I]),
    'cy_wino4s2_dgrad_packed_floats': (_L, [_I, _I]),
    'cy_wino2_dgrad_packed_floats': (_L, [_I, _I]),
    'cy_wino2_wgrad_ws_floats': (_L, [_I, _I, _I]),
    'cy_wino_wgrad_ws_floats': (_L, [_I, _I, _I]),
    'cy_conv_wgrad_ws_floats': (_L, [C.POINTER(ConvWgrad)]),
    'cy_conv_gemm_ws_floats': (_L, [C.POINTER(ConvGemm)]),
    'cy_routing_bwd_ws_floats': (_L, [C.POINTER(RoutingBwd)]),
    'cy_routing_fwd_ws_floats': (_L, [C.POINTER(RoutingFwd)]),
}
EXPORTS = sorted(list(_SIGS) + list(_RET))
ABI_VERSION = 5     # what the signatures above were written against (include/capsyolo_hip.h, csrc/error.cpp)

_lib = None


class HipExtensionError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises HipExtensionError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            'libcapsyolo_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C cs231-capsule-yolo-traffic-sign-detection_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.capsyolo_abi_version.restype = C.c_int
    have = lib.capsyolo_abi_version()
    if have != ABI_VERSION:     # same symbol names with other argument lists would be called with shifted arguments
        raise HipExtensionError('%s has ABI version %d, these bindings were written for %d: rebuild it '
                                '(make -C cs231-capsule-yolo-traffic-sign-detection_amd/csrc)' % (LIB_PATH, have, ABI_VERSION))
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, C.c_int
    for name, (restype, argtypes) in _RET.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, restype
    _lib = lib
    return lib


TRACE = None        # a list: every C-ABI call appends its entry-point name (launch census of tests / tools); None = off


def call(name, *args):
    """Call an int-returning entry point; non-zero -> HipExtensionError with the library's message."""
    lib = load()
    if TRACE is not None:
        TRACE.append(name)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.capsyolo_last_error()
        raise HipExtensionError('%s failed (code %d): %s' % (name, rc, msg.decode() if msg else '?'))


def query(name, *args):
    return getattr(load(), name)(*args)
