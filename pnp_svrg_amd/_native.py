"""ctypes binding of the C-ABI library (include/pnp_hip.h -> pnp_svrg_amd/lib/libpnp_hip.so).

There is no CPU fallback: if the library is missing or a call fails this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PNP_HIP_LIB') or os.path.join(_HERE, 'lib', 'libpnp_hip.so')   # PNP_HIP_LIB: A/B timing of kernel builds

F32, F64 = 0, 1

_vp = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/pnp_hip.h declares
SIGNATURES = {
    'pnp_version': (_i, []),
    'pnp_last_error': (ctypes.c_char_p, []),
    'pnp_csmri_plan_create': (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i]),
    'pnp_csmri_plan_destroy': (_i, [_vp]),
    'pnp_csmri_sel_from_indices': (_i, [_vp, _vp, _i, _vp, _vp]),
    'pnp_csmri_pack_mask': (_i, [_vp, _vp, _vp, _vp]),
    'pnp_csmri_draw_thresholds': (_i, [_vp, _vp, _i, ctypes.c_uint64, ctypes.c_uint32, _i, _vp, _vp, _vp, _vp]),
    'pnp_csmri_sel_from_thresholds': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'pnp_csmri_draw_minibatch': (_i, [_vp, _vp, _i, ctypes.c_uint64, ctypes.c_uint32, _vp, _vp, _vp]),
    'pnp_counter_add': (_i, [_vp, ctypes.c_uint32, _vp]),
    'pnp_log_append': (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    'pnp_log_append_inc': (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    'pnp_csmri_sel_from_dense': (_i, [_vp, _vp, _vp, _vp]),
    'pnp_csmri_pack_y': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'pnp_csmri_grad': (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _d, _vp, _vp, _vp]),
    'pnp_csmri_grad_sel': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _d, _vp, _d, _vp, _vp, _vp]),
    'pnp_csmri_svrg_step': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _d, _vp, _d, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp]),
    'pnp_csmri_svrg_outer_step': (_i, [_vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp]),
    'pnp_csmri_svrg_outer_iteration': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _d, _i, _d, _d, _vp, _vp, _i, _i, _vp, _vp]),
    'pnp_deblur_plan_create': (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    'pnp_deblur_plan_destroy': (_i, [_vp]),
    'pnp_deblur_grad': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _vp]),
    'pnp_deblur_grad_mb': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _vp]),
    'pnp_deblur_forward': (_i, [_vp, _vp, _vp, _vp]),
    'pnp_pr_workspace_elems': (_sz, [_i, _i]),
    'pnp_pr_grad': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _vp, _vp, _vp]),
    'pnp_pr_grad_batch': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp]),
    'pnp_pr_spectral_apply': (_i, [_vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp, _vp]),
    'pnp_sigma_est': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'pnp_prox_tv': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _d, _d, _vp, _vp, _vp, _vp]),
    'pnp_nlm2d': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _d, _d, _vp, _d, _vp, _vp, _vp, _vp]),
    'pnp_sse': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'pnp_minmax': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'pnp_dncnn_plan_create': (_i, [ctypes.POINTER(_vp), _i, _vp, _vp, _vp, _vp, _i, _i, _i]),
    'pnp_dncnn_plan_destroy': (_i, [_vp]),
    'pnp_dncnn_set_affine': (_i, [_vp, _vp, ctypes.c_float, ctypes.c_float]),
    'pnp_dncnn_set_winograd': (_i, [_vp, _i]),
    'pnp_dncnn_forward': (_i, [_vp, _vp, _vp, _vp]),
    'pnp_dncnn_denoise': (_i, [_vp, _vp, _vp, _i, _d, _vp, _vp, _vp]),
    'pnp_mmo_denoise': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'pnp_dncnn_profile_begin': (_i, [_vp, _i]),
    'pnp_dncnn_profile_end': (_i, [_vp, ctypes.POINTER(_d), ctypes.POINTER(ctypes.c_long)]),
    'pnp_dncnn_debug_clock': (_i, [_vp, _i, ctypes.POINTER(_d), ctypes.POINTER(_d), _vp]),
    'pnp_dncnn_debug_w44_floats': (_sz, []),
    'pnp_dncnn_debug_w44_weights': (_i, [_vp, _i, _vp, _vp]),
    'pnp_dncnn_debug_mid_layer': (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp]),
    'pnp_draw_thresholds': (_i, [_i, _i, _i, ctypes.c_uint64, ctypes.c_uint32, _i, _vp, _vp, _vp]),
    'pnp_indicator_from_thresholds': (_i, [_i, _i, _vp, _vp, _vp]),
    'pnp_rows_from_thresholds': (_i, [_i, _i, _i, _vp, _vp, _vp]),
    'pnp_indicator_from_indices': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'pnp_saga_table_update': (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _sz, _i, _vp]),
    'pnp_axpbypcz': (_i, [_d, _vp, _d, _vp, _d, _vp, _vp, _sz, _i, _vp]),
    'pnp_legacy_choice': (_i, [_vp, ctypes.POINTER(ctypes.c_int), _vp, _i, _i, _vp, _vp]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C pnp_svrg_amd/csrc`). There is no CPU fallback.')
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(status, what=''):
    if status != 0:
        msg = lib().pnp_last_error().decode(errors='replace')
        raise NativeError(f'{what} failed (status {status}): {msg}')


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
