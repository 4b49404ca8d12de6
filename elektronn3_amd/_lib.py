"""ctypes binding of libe3unet.so (include/e3unet.h).

This is the whole Python<->native boundary: plain pointers and sizes, no torch types cross it.  torch is
imported FIRST so that libe3unet.so resolves ``libamdhip64.so.7`` to the HIP runtime instance torch has already
loaded (same process-wide runtime => torch's streams and device pointers are valid inside the library).

There is no CPU fallback: if the library is missing and cannot be built, importing the ops raises.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# E3_LIB_PATH: developer A/B switch -- load another BUILD of the same library (e.g. the previous round's, kept under tools/_bin/) to time
# two versions of a kernel in one GPU session.  It must export the same ABI; it is never a fallback (a missing file raises like the default).
_LIB_PATH = os.environ.get('E3_LIB_PATH') or os.path.join(_HERE, 'libe3unet.so')

E3_FWD_TRAINING = 1
E3_FWD_SOFTMAX = 2
E3_FWD_FROZEN_BN = 4
E3_FWD_REUSE_PACKED = 8
E3_BWD_FROZEN_BN = 1


def E3_BWD_CU_RESERVE(n):
    """flag bits of e3_unet_backward2: CUs (multiple of 8, <= 128) left to a collective after the bucket event (include/e3unet.h)"""
    return ((int(n) >> 3) & 0x1f) << 8



class E3Error(RuntimeError):
    pass


class UNetCfg(ctypes.Structure):
    _fields_ = [('in_channels', c_int32), ('out_channels', c_int32), ('n_blocks', c_int32), ('start_filts', c_int32),
                ('planar_mask', c_uint32), ('normalization', c_int32), ('bn_eps', c_float), ('full_norm', c_int32), ('merge_add', c_int32), ('num_groups', c_int32), ('up_resize', c_int32), ('conv_valid', c_int32), ('act_slope', c_float), ('attention', c_int32), ('resunet', c_int32), ('enc_res_blocks', c_int32), ('dec_res_blocks', c_int32)]


class TileView(ctypes.Structure):
    """e3_tile_view of include/e3unet.h (a tile read in place from the padded volume, its kept region written in place: e3_unet_forward_tile)."""
    _fields_ = [('x', c_void_p), ('x_stride', ctypes.c_longlong * 3), ('y', c_void_p), ('y_stride', ctypes.c_longlong * 4)]


class CEDiceArgs(ctypes.Structure):
    """e3_ce_dice_args of include/e3unet.h (criterion evaluated inside the head, e3_unet_forward_loss)."""
    _fields_ = [('target', c_void_p), ('class_weight', c_void_p), ('ce_weight', c_float), ('dice_weight', c_float), ('eps', c_float), ('smooth', c_float),
                ('workspace', c_void_p), ('workspace_bytes', c_size_t), ('loss_out', c_void_p), ('sums_out', c_void_p)]


_P = c_void_p  # device pointer
_I = c_int
_F = c_float
_SIG = {
    'e3_last_error': (c_char_p, []),
    'e3_version': (c_char_p, []),
    'e3_unet_plan_create': (_I, [POINTER(UNetCfg), POINTER(c_void_p)]),
    'e3_unet_plan_destroy': (None, [c_void_p]),
    'e3_unet_param_count': (_I, [c_void_p]),
    'e3_unet_param_info': (_I, [c_void_p, _I, c_char_p, _I, POINTER(c_int64), POINTER(c_int)]),
    'e3_unet_bn_count': (_I, [c_void_p]),
    'e3_unet_out_dims': (_I, [c_void_p, _I, _I, _I, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'e3_unet_sizes': (_I, [c_void_p, _I, _I, _I, _I, _I, POINTER(c_size_t), POINTER(c_size_t)]),
    'e3_unet_forward': (_I, [c_void_p, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_float), _P,
                             _P, c_size_t, _P, c_size_t, c_uint32]),
    'e3_unet_forward_loss': (_I, [c_void_p, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_float), _P,
                                  _P, c_size_t, _P, c_size_t, c_uint32, POINTER(CEDiceArgs)]),
    'e3_unet_forward_roi': (_I, [c_void_p, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), _P, _P, c_size_t, c_uint32, POINTER(c_int)]),
    'e3_unet_forward_tile': (_I, [c_void_p, _P, POINTER(TileView), _I, _I, _I, _I, POINTER(c_void_p), _P, c_size_t, c_uint32, POINTER(c_int)]),
    'e3_unet_forward_roi_bf16': (_I, [c_void_p, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), _P, _P, c_size_t, c_uint32, POINTER(c_int)]),
    'e3_unet_forward_roi_f16': (_I, [c_void_p, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), _P, _P, c_size_t, c_uint32, POINTER(c_int)]),
    'e3_unet_backward': (_I, [c_void_p, _P, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_void_p), _P,
                              _P, c_size_t, _P, c_size_t, _P, _I]),
    'e3_unet_bf16_supported': (_I, [c_void_p]),
    'e3_unet_sizes_bf16': (_I, [c_void_p, _I, _I, _I, _I, _I, POINTER(c_size_t), POINTER(c_size_t)]),
    'e3_unet_forward_bf16': (_I, [c_void_p, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_float), _P,
                                  _P, c_size_t, _P, c_size_t, c_uint32]),
    'e3_unet_backward_bf16': (_I, [c_void_p, _P, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_void_p), _P,
                                   _P, c_size_t, _P, c_size_t, _P, _I]),
    'e3_conv3d_workspace_bytes_bf16': (c_size_t, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_conv3d_stats_parts_bf16': (_I, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_conv3d_fwd_bf16': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, c_size_t]),
    'e3_conv3d_dgrad_bf16': (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_conv3d_wgrad_bf16': (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_convT_workspace_bytes_bf16': (c_size_t, [_I, _I, _I, _I, _I, _I]),
    'e3_convT_stats_parts_bf16': (_I, [_I, _I, _I, _I, _I]),
    'e3_convT_fwd_bf16': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, c_size_t]),
    'e3_convT_dgrad_bf16': (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_convT_wgrad_bf16': (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_unet_f16_supported': (_I, [c_void_p]),
    'e3_unet_sizes_f16': (_I, [c_void_p, _I, _I, _I, _I, _I, POINTER(c_size_t), POINTER(c_size_t)]),
    'e3_unet_forward_f16': (_I, [c_void_p, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_float), _P,
                                  _P, c_size_t, _P, c_size_t, c_uint32]),
    'e3_unet_backward_f16': (_I, [c_void_p, _P, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_void_p), _P,
                                   _P, c_size_t, _P, c_size_t, _P, _I]),
    'e3_conv3d_workspace_bytes_f16': (c_size_t, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_conv3d_stats_parts_f16': (_I, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_conv3d_fwd_f16': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, c_size_t]),
    'e3_conv3d_dgrad_f16': (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_conv3d_wgrad_f16': (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_convT_workspace_bytes_f16': (c_size_t, [_I, _I, _I, _I, _I, _I]),
    'e3_convT_stats_parts_f16': (_I, [_I, _I, _I, _I, _I]),
    'e3_convT_fwd_f16': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, c_size_t]),
    'e3_convT_dgrad_f16': (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_convT_wgrad_f16': (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_unet_backward2': (_I, [c_void_p, _P, _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_void_p), _P,
                               _P, c_size_t, _P, c_size_t, _P, _I, c_uint32]),
    'e3_unet_backward_loss': (_I, [c_void_p, _P, _P, POINTER(CEDiceArgs), _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_void_p), _P,
                                   _P, c_size_t, _P, c_size_t, _P, _I, c_uint32]),
    'e3_unet_backward_loss_bf16': (_I, [c_void_p, _P, _P, POINTER(CEDiceArgs), _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_void_p), _P,
                                        _P, c_size_t, _P, c_size_t, _P, _I]),
    'e3_unet_backward_loss_f16': (_I, [c_void_p, _P, _P, POINTER(CEDiceArgs), _P, _P, _I, _I, _I, _I, POINTER(c_void_p), POINTER(c_void_p), _P,
                                       _P, c_size_t, _P, c_size_t, _P, _I]),
    'e3_unet_conv_count': (_I, [c_void_p]),
    'e3_unet_conv_info': (_I, [c_void_p, _I, c_char_p, _I, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'e3_unet_profile_select': (_I, [c_void_p, _I, _I]),
    'e3_unet_profile_read': (_I, [c_void_p, POINTER(c_double), POINTER(c_int)]),
    # per-op
    'e3_conv3d_workspace_bytes': (c_size_t, [_I, _I, _I]),
    'e3_conv3d_stats_parts': (_I, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_conv3d_fwd': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, c_size_t]),
    'e3_conv3d_dgrad': (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_conv3d_wgrad_workspace_bytes': (c_size_t, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_conv3d_wgrad': (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_convT_workspace_bytes': (c_size_t, [_I, _I, _I]),
    'e3_convT_stats_parts': (_I, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_convT_fwd': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, c_size_t]),
    'e3_convT_dgrad': (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_convT_wgrad_workspace_bytes': (c_size_t, [_I, _I, _I, _I, _I, _I, _I]),
    'e3_convT_wgrad': (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_bn_finalize': (_I, [_P, _P, _I, _I, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    'e3_bn_relu_apply': (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I]),
    'e3_maxpool': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I]),
    'e3_bn_bwd_workspace_bytes': (c_size_t, [_I, _I, _I, _I, _I]),
    'e3_bn_relu_bwd': (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I,
                            _P, _I, _P, _P, _P, _P, c_size_t]),
    'e3_conv1_fwd': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    'e3_conv1_bwd_workspace_bytes': (c_size_t, [_I, _I, _I, _I, _I, _I]),
    'e3_conv1_bwd': (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, c_size_t]),
    'e3_ce_dice_workspace_bytes': (c_size_t, [_I]),
    'e3_ce_dice_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P, c_size_t, _P]),
    'e3_ce_dice_sums': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, c_size_t, _P]),
    'e3_ce_dice_from_sums': (_I, [_P, _P, _P, _I, _F, _F, _F, _F, _P, c_size_t, _P]),
    'e3_ce_dice_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, c_size_t, _P, _P]),
    'e3_swa_update': (_I, [_P, _I, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int64]),
    'e3_swa_swap': (_I, [_P, _I, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64)]),
    'e3_unet_set_rrelu': (_I, [_P, c_double, c_double, c_uint32]),
    'e3_unet_attention_map': (_I, [c_void_p, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'e3_adamw_state_floats': (c_size_t, [_I, POINTER(c_int64)]),
    'e3_adamw_state_offset': (c_size_t, [_I, POINTER(c_int64), _I]),
    'e3_adamw_step': (_I, [_P, _I, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), _P, _P, _P, _P,
                           c_double, c_double, c_double, c_double, c_double, _P, _P]),
    'e3_adamw_step_bf16': (_I, [_P, _I, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), _P, _P, _P, _P,
                                c_double, c_double, c_double, c_double, c_double, _P, _P]),
    'e3_ncdhw_to_ndhwc': (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    'e3_ndhwc_to_ncdhw': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I]),
}

EXPORTED_SYMBOLS = tuple(_SIG)

_lib = None


def lib_path():
    return _LIB_PATH


def load(build_if_missing=True):
    """Load (building first if needed) libe3unet.so.  Raises E3Error if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and not os.environ.get('E3_LIB_PATH'):
        try:
            from .build import build
            build()
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(_LIB_PATH):
                raise E3Error(f'libe3unet.so is missing and could not be built: {e}') from e
    if not os.path.exists(_LIB_PATH):
        raise E3Error(f'{_LIB_PATH} not found: the HIP extension is required (there is no CPU fallback)')
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in _SIG.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().e3_last_error().decode('utf-8', 'replace')
        exc = {1: ValueError, 3: NotImplementedError}.get(rc, E3Error)
        raise exc(f'libe3unet error {rc}: {msg}')


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


__all__ = ['load', 'check', 'ptr', 'stream_ptr', 'UNetCfg', 'E3Error', 'EXPORTED_SYMBOLS', 'E3_FWD_TRAINING',
           'E3_FWD_SOFTMAX', 'E3_FWD_FROZEN_BN', 'E3_FWD_REUSE_PACKED', 'E3_BWD_FROZEN_BN', 'E3_BWD_CU_RESERVE', 'byref', 'c_size_t', 'c_void_p', 'c_float', 'c_int', 'c_int64', 'c_double', 'POINTER']
