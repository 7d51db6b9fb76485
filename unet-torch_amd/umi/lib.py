"""ctypes binding of libunetmi.so (declared in include/unetmi.h).

The product path has NO CPU or eager-PyTorch fallback: if the shared library is
missing and cannot be built, importing this module raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_uint, c_void_p

from . import build as _build

UMI_F32, UMI_F16 = 0, 1
CONV_UPSAMPLE2, CONV_FORCE_GENERIC, CONV_DGRAD_STRIDED, CONV_ACCUMULATE = 1, 2, 4, 8

_ERR = {-1: "UMI_ERR_BADARG", -2: "UMI_ERR_UNSUPPORTED", -3: "UMI_ERR_WORKSPACE"}


def _load():
    alt = os.environ.get("UMI_LIB_OVERRIDE")          # tuning aid: A/B two builds of libunetmi on one box
    if alt:
        if not os.path.exists(alt):
            raise RuntimeError(f"UMI_LIB_OVERRIDE={alt} does not exist")
        return ctypes.CDLL(alt)
    path = _build.LIB
    if not os.path.exists(path) or (_build.stale() and os.path.exists(_build.HIPCC)):
        if not os.path.exists(_build.HIPCC):
            raise RuntimeError(
                f"libunetmi.so not found at {path} and hipcc is unavailable: the HIP extension is "
                "required, there is no CPU fallback (run `python __graft_entry__.py build`).")
        _build.build_lib()
    return ctypes.CDLL(path)


_lib = _load()

# name -> (restype, argtypes); mirrors include/unetmi.h one to one
SIGNATURES = {
    "umi_version": (c_int, []),
    "umi_arch": (c_char_p, []),
    "umi_tune_conv3x3_impl": (c_int, [c_int]),
    "umi_zoom_cubic_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "umi_zoom_cubic_hwc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "umi_linear_fused": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_int, c_float,
                                 c_uint, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "umi_pack_kn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long, c_int, c_int, c_int,
                            c_int, c_void_p]),
    "umi_pack_kn8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long, c_int, c_int, c_int,
                             c_int, c_void_p]),
    "umi_conv_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_conv_fwd_plan": (c_int, [c_int] * 15 + [c_void_p, c_void_p]),
    "umi_bn_finalize": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_float, c_float,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "umi_pool2_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_void_p]),
    "umi_pool2_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                              c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_bn_bwd_ws_bytes": (c_size_t, [c_long, c_int]),
    "umi_bn_bwd_reduce": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "umi_bn_bwd_apply": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_long, c_int, c_int, c_void_p]),
    "umi_conv_wgrad_ws_bytes": (c_size_t, [c_int] * 9),
    "umi_conv_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                               c_long, c_long, c_long, c_float,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "umi_conv_wgrad_deferred": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                        c_long, c_long, c_long, c_float,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "umi_wgrad_reduce_group": (c_int, [c_int, c_void_p, c_void_p]),
    "umi_conv_wgrad_group": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_long, c_float, c_long,
                                     c_int, c_int, c_int, c_void_p]),
    "umi_conv_wgrad_bnapply": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_void_p, c_long, c_long, c_long, c_float,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p, c_size_t, c_void_p]),
    "umi_colsum_ws_bytes": (c_size_t, [c_long, c_int]),
    "umi_colsum_group": (c_int, [c_int, c_void_p, c_int, c_void_p, c_float, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "umi_materialize_nchw": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_wstd_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "umi_wstd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "umi_wstd_fwd_multi": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "umi_wstd_bwd_multi": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "umi_gn_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                           c_int, c_long, c_int, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "umi_gn_fwd_ws_bytes": (c_size_t, [c_int, c_long, c_int]),
    "umi_gn_bwd_ws_bytes": (c_size_t, [c_int, c_long, c_int, c_int]),
    "umi_gn_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                           c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_int, c_long, c_int, c_int, c_int,
                           c_void_p, c_size_t, c_void_p, c_void_p]),
    "umi_gn_param_grads_group": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p]),
    "umi_pool3s2_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_pool3s2_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_void_p]),
    "umi_ln_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_long, c_int, c_float,
                           c_int, c_void_p]),
    "umi_ln_bwd_ws_bytes": (c_size_t, [c_long, c_int]),
    "umi_ln_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                           c_float, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "umi_elementwise": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_long, c_int, c_void_p]),
    "umi_dropout": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_float, ctypes.c_uint, c_long, c_int, c_int,
                            c_void_p, c_void_p, c_void_p]),
    "umi_dropout_fused": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_float, ctypes.c_uint, c_long, c_int, c_int,
                                  c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "umi_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                             c_void_p]),
    "umi_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_bilinear2x": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_colsum": (c_int, [c_void_p, c_int, c_void_p, c_float, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "umi_conv3x3_fwd_act": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p]),
    "umi_conv_dgrad_bnred": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_conv_wgrad_bias": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_long, c_long, c_long, c_void_p, c_float]
                            + [c_int] * 9 + [c_void_p, c_size_t, c_void_p, c_void_p]),
    "umi_conv_gather_bnred_rows": (c_int, [c_int] * 15),
    "umi_conv_gather_bnred": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
                              + [c_int] * 13 + [c_void_p]),
    "umi_head_bwd_fused_ws_bytes": (c_size_t, [c_long, c_int, c_int]),
    "umi_head_bwd_fused": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_long, c_long, c_float, c_void_p, c_size_t, c_long, c_int, c_int, c_int, c_void_p]),
    "umi_head_dgrad_bnred_rows": (c_int, [c_long, c_int, c_int, c_int, c_int]),
    "umi_head_dgrad_bnred": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_long, c_int, c_int, c_int, c_void_p]),
    "umi_bn_bwd_from_partials": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "umi_bn_stats_rows": (c_int, [c_long, c_int]),
    "umi_bn_stats": (c_int, [c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_void_p]),
    "umi_pool2_bwd_bnred_stat_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "umi_pool2_bwd_bnred": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                    c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "umi_dice_ce_ws_bytes": (c_size_t, [c_int, c_int, c_long]),
    "umi_dice_ce_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umi_dice_ce_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p]),
    "umi_optim_block_elems": (c_int, []),
    "umi_table_upload": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "umi_optim_sgd_multi": (c_int, [c_void_p, c_int, c_int, c_double, c_double, c_double, c_double, c_int, c_int, c_void_p]),
    "umi_optim_adam_multi": (c_int, [c_void_p, c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_double,
                                     c_void_p]),
    "umi_optim_hyper_bytes": (c_size_t, []),
    "umi_optim_hyper_pre": (c_int, [c_void_p, c_int, c_void_p]),
    "umi_optim_hyper_poly": (c_int, [c_void_p, c_void_p]),
    "umi_optim_sgd_multi_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_double, c_double, c_double, c_int, c_int, c_void_p]),
    "umi_optim_adam_multi_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_double, c_double, c_double, c_double, c_void_p]),
    "umi_pack_block_elems": (c_int, []),
    "umi_pack_kn_multi": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "umi_add2_relu_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_long, c_int, c_int,
                                  c_void_p]),
    "umi_add2_relu_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_int,
                                  c_void_p]),
    "umi_gate_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p]),
    "umi_gate_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                             c_long, c_int, c_int, c_void_p]),
    "umi_znorm_ws_bytes": (c_size_t, []),
    "umi_znorm_hwc": (c_int, [c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "umi_argmax_mask": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_void_p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(_lib, _name)          # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"libunetmi: {what} failed with "
                           f"{_ERR.get(status, 'hipError_t ' + str(status))}")


def fn(name):
    return getattr(_lib, name)


lib = _lib
