"""ctypes binding of libkp_hip.so (the C ABI declared in include/kp_hip.h).

There is deliberately no fallback: if the shared library is missing or fails to
load, importing any compute entry point raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C keypoints_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libkp_hip.so')

KP_F32, KP_BF16, KP_F16 = 0, 1, 2
TAIL_NONE, TAIL_MAXPOOL2, TAIL_UPSAMPLE2 = 0, 1, 2
KP_CFG_CUS = 6   # kp_config key: CU budget of the persistent kernels (include/kp_hip.h)
COMBINE = {'sum_and_clamp': 1, 'pretrained_network': 2, 'loop': 3}

_p, _i, _f, _z, _ll = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong

# name -> argument ctypes (all return int unless noted); mirrors include/kp_hip.h one to one
SIGNATURES = {
    'kp_nchw_to_nhwc': [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_nhwc_grad_to_nchw': [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_nchw2_to_nhwc': [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_nhwc_grad_to_nchw2': [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_first_stats': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_first_fwd_bnact': [_p, _p, _p, _p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_first_bwd_reduce': [_p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_first_bwd_wgrad': [_p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_pad_hw': [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_fwd_act': [_p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_pad_border': [_p, _i, _i, _i, _i, _i, _i, _p],
    'kp_pack_weights': [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    'kp_pack_weights_batch': [_i, _p, _p, _p, _p, _p, _p, _i, _p],
    'kp_wgrad_unpack': [_p, _i, _p, _i, _i, _i, _i, _p],
    'kp_conv_fwd': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_dgrad': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_wgrad': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_bn_finalize': [_p, _ll, _p, _p, _p, _p, _p, _p, _f, _f, _ll, _i, _p],
    'kp_bn_act_fwd': [_p, _p, _p, _p, _p, _f, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    'kp_bn_act_bwd_reduce': [_p, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _p],
    'kp_bn_bwd_finalize': [_p, _p, _ll, _i, _p, _p, _p, _p],
    'kp_bn_act_bwd_apply': [_p, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_upsample2_bwd': [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    'kp_head_fwd': [_p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_head_bwd': [_p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_dgrad_bnred': [_p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_head_fwd_bnin': [_p, _p, _p, _p, _p, _f, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_head_bwd_bnin': [_p, _p, _p, _p, _p, _p, _p, _f, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_head_fwd_t': [_p, _i, _p, _p, _p, _p, _f, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_head_bwd_t': [_p, _p, _p, _i, _p, _p, _p, _p, _f, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_keypoint_bottleneck_fwd_t': [_p] * 12 + [_i] * 5 + [_f, _f, _i, _p],
    'kp_keypoint_bottleneck_bwd_t': [_p] * 13 + [_i] * 5 + [_f, _f, _i, _p],
    'kp_nhwc_to_nchw': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_fwd_pair': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_fwd_bnin_pair': [_p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_bn_finalize_pair': [_p, _ll, _p, _p, _p, _p, _p, _f, _f, _ll, _i, _p],
    'kp_head_fwd_bnin_pair': [_p, _p, _p, _p, _p, _f, _p, _p, _f, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_unit_infer_pack': [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _p],
    'kp_unit_infer_keypoints_fwd': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    'kp_unit_infer_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    'kp_spatial_softmax_fwd': [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    'kp_spatial_softmax_bwd': [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    'kp_gaussian_fwd': [_p, _p, _i, _i, _i, _f, _f, _p],
    'kp_gaussian_bwd': [_p, _p, _p, _i, _i, _i, _f, _f, _p],
    'kp_point_map_fwd': [_p, _p, _i, _i, _i, _p],
    'kp_transport_max_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'kp_transport_max_bwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'kp_transport_fwd': [_i] + [_p] * 7 + [_i] * 5 + [_p],
    'kp_transport_bwd': [_i] + [_p] * 9 + [_i] * 5 + [_p],
    'kp_keypoint_bottleneck_fwd': [_p] * 12 + [_i] * 5 + [_f, _f, _p],
    'kp_keypoint_bottleneck_bwd': [_p] * 13 + [_i] * 5 + [_f, _f, _p],
    'kp_warp_tps': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_warp_affine': [_p, _p, _p, _i, _i, _i, _i, _p],
    'kp_warp_rotate': [_p, _p, _p, _i, _i, _i, _i, _p],
    'kp_l2_loss': [_p, _p, _p, _p, _p, _f, _z, _p],
    'kp_l2_loss_mean': [_p, _p, _p, _p, _p, _p, _z, _p],
    'kp_scale': [_p, _p, _p, _z, _p],
    'kp_adam_step': [_p, _p, _p, _p, _z, _f, _f, _f, _f, _f, _f, _p],
    'kp_amp_init': [_p, C.c_double, _p],
    'kp_amp_check': [_p, _z, _p, _p],
    'kp_amp_scale': [_p, _p, _p, _z, _p],
    'kp_amp_prepare': [_p, _f, _f, _p],
    'kp_adam_step_scaled': [_p, _p, _p, _p, _z, _f, _f, _f, _f, _p, _p],
    'kp_amp_update': [_p, _f, _f, _i, _f, _p],
    'kp_conv_fwd_fin': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _f, _f, _ll, _p, _p],
    'kp_conv_fwd_bnin': [_p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'kp_conv_wgrad_bnin': [_p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'kp_comm_unique_id': [_p],
    'kp_comm_init': [_p, _i, _i, _p],
    'kp_comm_allreduce': [_p, _p, _z, _i, _i, _p],
    'kp_comm_broadcast': [_p, _p, _z, _i, _i, _p],
    'kp_comm_count': [_p, _p],
    'kp_comm_destroy': [_p],
    'kp_prof_enable': [_i],
    'kp_config': [_i, _i],
    'kp_config_ptr': [_i, _p],
    'kp_prof_collect': [_i, _p, _p, _p],
    'kp_prof_bytes': [_i, _p],
    'kp_prof_step_bytes': [_p, _p],
}
PLAIN = {'kp_config_get': ([_i], _i), 'kp_conv_fwd_act_supported': ([_i] * 6, _i), 'kp_conv_fwd_fin_folds': ([_i] * 7, _i), 'kp_comm_available': ([], _i), 'kp_conv_pair_supported': ([_i] * 7, _i), 'kp_head_pair_supported': ([_i] * 7, _i), 'kp_conv_wgrad_pad_hw': ([_i] * 7 + [_p, _p], _i), 'kp_conv_first_fused_supported': ([_i] * 6, _i), 'kp_conv_first_fused_rows': ([_i] * 6, _ll), 'kp_bn_finalize_pair_supported': ([_ll, _i], _i), 'kp_head_t_supported': ([_i] * 6, _i), 'kp_keypoint_bottleneck_t_supported': ([_i] * 6, _i), 'kp_conv_bnin_supported': ([_i] * 6, _i), 'kp_head_bnin_supported': ([_i] * 6, _i), 'kp_conv_dgrad_bnred_supported': ([_i] * 6, _i), 'kp_conv_dgrad_bnred_rows': ([_i] * 6, _ll), 'kp_unit_infer_param_floats': ([_i, _i, _p], _ll), 'kp_conv_wgrad_partials': ([_i] * 8, _i), 'kp_conv_fwd_stats_rows': ([_i] * 7, _ll), 'kp_bn_bwd_rows': ([_i] * 7, _ll), 'kp_head_bwd_ws_floats': ([_i] * 6, _ll), 'kp_last_rows': ([], _ll), 'kp_bn_ws_doubles': ([_i], _i), 'kp_version': ([], _i), 'kp_last_error': ([], C.c_char_p), 'kp_device_cus': ([], _i)}

_lib = None


class KpError(RuntimeError):
    rc = None   # the KP_E_* code of a failed call (None: raised by the loader itself)


def lib():
    """The loaded CDLL (loads on first use; raises if libkp_hip.so is absent)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KpError(f'{LIB_PATH} not found: build the HIP extension first '
                          '(make -C keypoints_amd/csrc); there is no CPU fallback')
        _lib = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, _i
        for name, (args, res) in PLAIN.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, res
    return _lib


def call(name: str, *args):
    """Invoke a kp_* entry point; non-zero return -> KpError with the library's message."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        e = KpError(f'{name} failed ({rc}): {lib().kp_last_error().decode()}')
        e.rc = rc
        raise e


KP_E_UNSUPPORTED = -3


def try_call(name: str, *args) -> bool:
    """`call`, but a KP_E_UNSUPPORTED answer (the library declined the shape / alignment / configuration and launched nothing)
    is returned as False instead of raised: the caller then runs the equivalent unfused sequence."""
    rc = getattr(lib(), name)(*args)
    if rc == KP_E_UNSUPPORTED:
        return False
    if rc != 0:
        e = KpError(f'{name} failed ({rc}): {lib().kp_last_error().decode()}')
        e.rc = rc
        raise e
    return True


def last_rows() -> int:
    """Rows of the partial-sum buffer written by the latest kp_conv_fwd(stats) / kp_bn_act_bwd_reduce call."""
    return int(lib().kp_last_rows())


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
