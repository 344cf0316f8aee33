"""ctypes binding of the C-ABI library ``lib/libsmsut_hip.so`` (declared in ``include/smsut_hip.h``).

PyTorch is used here only for device memory and the current HIP stream.  There is NO fallback:
if the library is missing or a tensor is not on a HIP device the call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsmsut_hip.so")

# signature codes: p = device pointer, i = int32, l = int64, f = float, d = double, s = stream (void*)
# return type is int (0 = ok) unless the name is listed in _RET_I64.
SIGNATURES: Dict[str, str] = {
    # norm.hip
    "smsut_in_chunks": "iii",
    "smsut_instnorm_fwd": "ppppppp iii ff i s",
    "smsut_instnorm_fwd_partials": "ppppppp iiii ff i s",
    "smsut_instnorm_fwd_partials_hs": "ppppppp iiii ff i s",
    "smsut_instnorm_fwd_partials_hs2": "ppppppp iiii ff i s",
    "smsut_in_finalize_fwd": "p i pp iii f s",
    "smsut_in_finalize_fwd2": "p i pp p i pp iii f s",
    "smsut_in_finalize_bwd": "p i pp iii s",
    "smsut_in_apply_bwd": "pppppppp pp iii s",
    "smsut_restail_fwd": "pppppppppp p iii f s",
    "smsut_restail_fwd_hs": "pppppppppp p iii f s",
    "smsut_restail_bwd_hs": "pppppppppppp pp ppp pppp pp iii f s",
    "smsut_in_apply_bwd_hs": "pppppppp pp p iii s",
    "smsut_restail_bwd": "pppppppppppp pp ppp pppp p iii f s",
    "smsut_restail_bwd_fin": "pppppppppppp pp ppp pppp pp iii f s",
    "smsut_restail_bwd_amax": "pppppppppppp pp ppp pppp pp iii f s",
    "smsut_in_apply_bwd_amax": "pppppppp pp p iii s",
    "smsut_absmax_finish": "p i p s",
    "smsut_amax_blocks": "iii",
    "smsut_instnorm_bwd": "pppppp ppppp p iii f s",
    "smsut_sgd_momentum_multi": "ppp i fff s",
    "smsut_sgd_chunk": "",
    "smsut_restail_fwd_pool": "pppppppppp ppp iiii f i s",
    "smsut_restail_bwd_pool": "ppp pppppppppp pp ppp pppp p pp iiii f i s",
    "smsut_instnorm_pool_fwd_partials": "ppppppp iiiii ff s",
    "smsut_instnorm_pool_bwd": "pppppp ppppp p iiii f s",
    "smsut_instnorm_bwd2": "ppppppppppp ppp pp iii f s",
    # conv_naive.hip
    "smsut_conv2d_fwd_generic": "pppp iiiiiiiiiii s",
    "smsut_conv2d_dgrad_generic": "ppp iiiiiiiiiii s",
    "smsut_conv2d_wgrad_generic_ws": "iiiiiii",
    "smsut_conv2d_wgrad_generic": "pppp iiiiiiiiiii s",
    "smsut_colsum_ws": "li",
    "smsut_colsum": "ppp li s",
    # conv_mfma.hip
    "smsut_conv2d_mfma_supported": "iiiii",
    "smsut_conv2d_fwd_mfma": "ppp iiiiii i s",
    "smsut_conv2d_fwd_mfma_pre": "ppp iiiiii i p s",
    "smsut_conv2d_mfma_tiles": "iiiiiii",
    "smsut_conv2d_mfma_persistent": "iiiiiii",
    "smsut_conv2d_dgrad_mfma_bwdstats": "ppppppppp f iiiii s",
    "smsut_conv2d_dgrad_mfma_bwdstats_pre": "ppppppppp f iiiii p s",
    "smsut_conv2d_fwd_mfma_stats": "pppp iiiiii s",
    "smsut_conv2d_fwd_mfma_stats_pre": "pppp iiiiii p s",
    "smsut_conv2d_fwd_mfma_cfg": "ppp iiiiii ii s",
    "smsut_wino_image_floats": "ii",
    "smsut_wino_prepare": "ppppp i s",
    "smsut_conv2d_wgrad_mfma_supported": "iiiii",
    "smsut_conv2d_wgrad_mfma_ws": "iiiiii",
    "smsut_conv2d_wgrad_mfma": "pppp iiiiii s",
    "smsut_convT2x2_mfma_supported": "ii",
    "smsut_convT2x2_fwd_mfma": "ppp iiiii s",
    "smsut_convT2x2_dgrad_mfma": "ppp iiiii s",
    "smsut_convT2x2_wgrad_mfma_ws": "iiiii",
    "smsut_convT2x2_wgrad_mfma": "pppp iiiii s",
    "smsut_convT2x2_ps_supported": "ii",
    "smsut_convT2x2_fwd_ps": "ppp iiiii s",
    "smsut_convT2x2_wgrad_ps_ws": "iiiii",
    "smsut_convT2x2_wgrad_ps": "pppp iiiii s",
    # conv1x1.hip
    "smsut_conv1x1_supported": "ii",
    "smsut_conv1x1_tiles": "iii",
    "smsut_conv1x1_fwd": "pppp iiii i s",
    "smsut_conv1x1_wgrad_ws": "iiii",
    "smsut_conv1x1_wgrad": "pppp iiii s",
    "smsut_conv2d_fwd_mfma_stats_inaff": "pppp pppp f iiiii s",
    "smsut_conv2d_fwd_mfma_stats_inaff_pre": "pppp pppp f iiiii p s",
    "smsut_conv2d_wgrad_mfma_inaff": "pppp pppp f iiiii s",
    "smsut_conv2d_wgrad_mfma_slabs": "ppp pppp f iiiii s",
    "smsut_conv2d_mfma_form": "iiiiii",
    "smsut_conv2d_mfma_cat_supported": "iiiii",
    "smsut_conv2d_fwd_mfma_stats_cat": "ppppp iiiii s",
    "smsut_conv2d_fwd_mfma_stats_cat_pre": "ppppp iiiii p s",
    "smsut_conv2d_fwd_sc_supported": "iiiiii",
    "smsut_conv2d_fwd_mfma_stats_sc": "pppppppp iiiii s",
    "smsut_conv2d_fwd_mfma_stats_sc_f16": "pppppppp iiiii s",
    "smsut_conv2d_fwd_sc_f16_supported": "iiiiii",
    "smsut_conv2d_fwd_mfma_stats_sc_pre": "pppppppp iiiii p s",
    "smsut_conv2d_dgrad_sc_supported": "iiiiii",
    "smsut_conv2d_dgrad_mfma_sc": "pppppp iiiiii s",
    "smsut_conv2d_wgrad_sc_supported": "iiiii",
    "smsut_conv2d_wgrad_sc_ws": "iiiii",
    "smsut_conv2d_wgrad_mfma_sc": "pp i pppp iiiii s",
    "smsut_conv2d_wgrad_mfma_cat": "pp i ppp iiiiii s",
    "smsut_conv2d_fwd_mfma_stats_sc_fin": "pppppppp ppppp f iiiii p s",
    "smsut_conv2d_fwd_mfma_stats_inaff_fin": "pppppppp f ppp f iiiii p s",
    "smsut_conv2d_dgrad_mfma_bwdstats_fin": "ppppppppp f ppp iiiii p s",
    "smsut_conv2d_wgrad_pair_supported": "iiiiiiiii",
    "smsut_conv2d_wgrad_pair_ws": "iiiiiiiii",
    "smsut_conv2d_wgrad_pair": "pppppp i pppppp i i pp f pp iiii s",
    "smsut_conv2d_wgrad_pair_slabs": "pppp i pppp i pp f p iiii s",
    # 4x4 s1 p1 (networks.NLayerDiscriminator)
    "smsut_conv2d_k4_supported": "ii",
    "smsut_conv2d_k4_fwd": "ppp iiiii i s",
    "smsut_conv2d_k4_wgrad_ws": "iiiii",
    "smsut_conv2d_k4_wgrad": "pppp iiiii s",
    # fp16-operand forms (config 5)
    "smsut_conv2d_f16_supported": "iii",
    "smsut_conv2d_fwd_mfma_f16": "pppp iiiiii i s",
    "smsut_conv2d_fwd_mfma_stats_f16": "pppp iiiiii s",
    "smsut_conv2d_fwd_mfma_stats_cat_f16": "ppppp iiiii s",
    "smsut_conv2d_fwd_mfma_split_f16": "ppppp iiiiiii s",
    "smsut_conv2d_dgrad_mfma_bwdstats_f16": "pppppppppp f iiiii s",
    "smsut_conv2d_dgrad_mfma_bwdstats_f16_hs": "pppppppppp f iiiii s",
    "smsut_conv2d_f16_hs_supported": "iiiiii",
    "smsut_conv2d_fwd_mfma_stats_f16_hs": "ppppp iiiii s",
    "smsut_conv2d_fwd_mfma_stats_f16_hsx": "pppp iiiii s",
    "smsut_conv2d_wgrad_f16_xh": "ppppp iiiii s",
    "smsut_conv2d_wgrad_f16_xh_inaff": "ppppp pppp f iiiii s",
    "smsut_conv2d_fwd_mfma_stats_inaff_f16_hsx": "pppp pppp f iiiii s",
    "smsut_conv2d_fwd_mfma_stats_sc_f16_hs": "pppppppp iiiii s",
    "smsut_conv2d_wgrad_f16_supported": "iiiii",
    "smsut_conv2d_wgrad_f16_ws": "iiiii",
    "smsut_conv2d_wgrad_f16": "pp i pppp iiiii s",
    "smsut_absmax_scale_ws": "l",
    "smsut_absmax_scale": "p l pp s",
    "smsut_absmax_scale2": "p l p l pp s",
    "smsut_conv2d_dgrad_sc_f16_supported": "iiiiii",
    "smsut_conv2d_dgrad_mfma_sc_f16": "ppppppp iiiiii s",
    "smsut_conv2d_wgrad_sc_f16_supported": "iiiii",
    "smsut_conv2d_wgrad_sc_f16_ws": "iiiii",
    "smsut_conv2d_wgrad_sc_f16": "pp i ppppp iiiii s",
    "smsut_conv1x1_fwd_cat": "pp i ppp iiii s",
    "smsut_conv1x1_wgrad_cat": "pp i ppp iiii s",
    "smsut_conv2d_mfma_split_supported": "iiiiii",
    "smsut_conv2d_fwd_mfma_split": "pppp iiiiiii s",
    "smsut_conv2d_fwd_mfma_split_pre": "pppp iiiiiii p s",
    "smsut_conv1x1_fwd_split": "pppp iiiiii s",
    "smsut_conv1x1_thin_supported": "ii",
    "smsut_conv1x1_thin_dgrad": "ppp iiii s",
    "smsut_conv1x1_thin_wgrad_ws": "i",
    "smsut_conv1x1_thin_wgrad": "pppp iiii s",
    # conv_small.hip
    "smsut_conv2d_small_supported": "iii",
    "smsut_conv2d_small_fwd": "pppp iiiiiiiiii s",
    "smsut_conv2d_small_dgrad": "ppp iiiiiiiiii s",
    "smsut_conv2d_flat_wgrad_supported": "iiii",
    "smsut_conv2d_flat_wgrad_ws": "iiiiii",
    "smsut_conv2d_flat_wgrad": "pppp iiiiiiiiii s",
    # pointwise.hip
    "smsut_add_act": "ppp l f s",
    "smsut_act_bwd": "ppp l f s",
    "smsut_tanh_fwd": "pp l s",
    "smsut_tanh_bwd": "ppp l s",
    "smsut_bias_add": "ppp l i s",
    "smsut_row_lerp": "pppp ll s",
    "smsut_fill": "p f l s",
    "smsut_scale": "pp f p l s",
    "smsut_maxpool2_fwd": "pp iiii s",
    "smsut_maxpool2_bwd": "ppp iiii s",
    "smsut_maxpool2_bwd_add": "pppp iiii s",
    "smsut_avgpool2_fwd": "pp iiii s",
    "smsut_avgpool2_bwd": "pp iiii s",
    "smsut_bilinear2_fwd": "pp iiii s",
    "smsut_warp_joint": "pppppp iiiiii s",
    "smsut_elastic_deform": "ppppp iiii s",
    "smsut_bilinear2_bwd": "pp iiii s",
    "smsut_window_fwd": "pp iiiiiiiii s",
    "smsut_window_bwd": "pp iiiiiiiii s",
    "smsut_blurdown_fwd": "pp iiii s",
    "smsut_blurdown_bwd": "pp iiii s",
    "smsut_copy_channels": "p ii p ii i l s",
    "smsut_concat2": "p i p i p l i s",
    "smsut_modal_planes": "ppp i l ii s",
    # loss.hip
    "smsut_dicece_ws": "i l ii",
    "smsut_dicece_stats": "ppppp i l ii s",
    "smsut_dicece_final": "ppp ii d ff s",
    "smsut_dicece_bwd": "ppppp i l ii d ff s",
    "smsut_sum_ws": "l i",
    "smsut_sum": "ppp l d s",
    "smsut_l1_fwd": "pppp l s",
    "smsut_l1_bwd": "ppppp l s",
    "smsut_softmax_mse_fwd": "pppp l i s",
    "smsut_softmax_mse_bwd": "pppp l i s",
    "smsut_argmax_channels": "pp l i s",
    "smsut_gp_fwd": "pppp i l s",
    "smsut_gp_bwd": "pppp i l s",
    "smsut_ce_rows_fwd": "ppp ii s",
    "smsut_ce_rows_bwd": "pppp ii s",
    "smsut_gather_rows": "ppp i l ii s",
    "smsut_scatter_rows": "ppp i l ii s",
    "smsut_l2norm_fwd": "ppp ii s",
    "smsut_l2norm_bwd": "pppp ii s",
    "smsut_patchnce_fwd": "pppp iii f s",
    "smsut_patchnce_bwd": "pppp iii f s",
}
_RET_I64 = {"smsut_wino_image_floats", "smsut_conv2d_wgrad_pair_ws", "smsut_convT2x2_wgrad_ps_ws", "smsut_conv2d_wgrad_sc_ws", "smsut_conv2d_k4_wgrad_ws", "smsut_conv2d_wgrad_f16_ws", "smsut_conv2d_wgrad_sc_f16_ws", "smsut_absmax_scale_ws", "smsut_conv2d_wgrad_generic_ws", "smsut_colsum_ws", "smsut_dicece_ws", "smsut_sum_ws",
            "smsut_conv2d_wgrad_mfma_ws", "smsut_convT2x2_wgrad_mfma_ws", "smsut_conv2d_flat_wgrad_ws", "smsut_conv1x1_wgrad_ws",
            "smsut_conv1x1_thin_wgrad_ws"}
_NO_STATUS = _RET_I64 | {"smsut_conv2d_k4_supported", "smsut_conv2d_f16_supported", "smsut_conv2d_wgrad_f16_supported", "smsut_in_chunks", "smsut_amax_blocks", "smsut_conv2d_mfma_supported", "smsut_conv2d_wgrad_mfma_supported",
                         "smsut_convT2x2_mfma_supported", "smsut_conv2d_small_supported",
                         "smsut_conv2d_flat_wgrad_supported", "smsut_conv2d_mfma_tiles", "smsut_conv2d_mfma_persistent", "smsut_conv1x1_supported",
                         "smsut_conv1x1_tiles", "smsut_conv1x1_thin_supported", "smsut_conv2d_mfma_split_supported", "smsut_conv2d_mfma_cat_supported",
                         "smsut_conv2d_fwd_sc_supported", "smsut_conv2d_fwd_sc_f16_supported", "smsut_conv2d_dgrad_sc_supported",
                         "smsut_conv2d_dgrad_sc_f16_supported", "smsut_conv2d_wgrad_sc_f16_supported", "smsut_conv2d_f16_hs_supported",
                         "smsut_conv2d_wgrad_sc_supported", "smsut_convT2x2_ps_supported", "smsut_conv2d_wgrad_pair_supported",
                         "smsut_conv2d_wgrad_mfma_slabs", "smsut_conv2d_wgrad_pair_slabs", "smsut_conv2d_mfma_form", "smsut_sgd_chunk"}     # (return a count / a form id, not a status)

_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_int64, "f": ctypes.c_float, "d": ctypes.c_double,
       "s": ctypes.c_void_p}

_lib: Optional[ctypes.CDLL] = None


class SmsutHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """dlopen the C-ABI library and bind every entry point; raises loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmsutHipError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = [_CT[c] for c in sig.replace(" ", "")]
        fn.restype = ctypes.c_int64 if name in _RET_I64 else ctypes.c_int
    _lib = lib
    return lib


def stream_ptr() -> int:
    if not torch.cuda.is_available():
        raise SmsutHipError("SMSUT HIP ops need an MI355X (no HIP device visible; there is no CPU fallback)")
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise SmsutHipError("SMSUT HIP ops need tensors on a HIP device (no CPU fallback in the product path)")
    return t.data_ptr()


def call(name: str, *args):
    """Invoke an entry point; tensors are passed as device pointers, the stream is appended by the caller."""
    lib = load()
    conv = [ptr(a) if isinstance(a, torch.Tensor) or a is None else a for a in args]
    rc = getattr(lib, name)(*conv)
    if name in _NO_STATUS:
        return rc
    if rc != 0:
        raise SmsutHipError(f"{name} failed with status {rc}" + (" (invalid argument)" if rc == -1 else " (hipError_t)"))
    return rc
