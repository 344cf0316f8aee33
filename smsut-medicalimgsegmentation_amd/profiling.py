"""Per-shape replay profile of one training step (measurement tooling, used by ``bench.py`` and ``scratch/``).

rocprofv3's per-kernel-name averages hide shapes (one template instantiation serves a dozen layer sizes), so the step is
profiled from the C-ABI side instead: every ``_hip.call`` of ONE eager step is recorded (entry point + converted
arguments), then each distinct (entry point, integer/float arguments) is replayed alone between HIP events on the
pointers it ran with (still owned by torch's caching allocator -- the replay happens before anything is freed back to
the driver).  Convolution-like entry points get their algorithmic FLOPs from the integer arguments
(2 * N * Ho * Wo * Cin * Cout * k^2), which gives

    step_conv_frac = sum(conv FLOPs of the step) / sum(conv kernel time of the step) / fp32 MFMA peak

-- the roofline fraction of the WHOLE conv side of the step, not of one hand-picked instantiation.
"""
from __future__ import annotations

import collections
from typing import Callable, Dict, List, Tuple

import torch

from . import _hip as H


def _k2(ints, i):
    return ints[i] * ints[i]


# entry point -> (flops(ints), kind); ``ints`` = the call's integer arguments in declaration order (include/smsut_hip.h)
_CONV_FLOPS: Dict[str, Tuple[Callable[[List[int]], float], str]] = {
    "smsut_conv2d_fwd_generic": (lambda a: 2.0 * a[0] * a[4] * a[5] * a[3] * a[6] * a[7] * a[8], "direct"),
    "smsut_conv2d_dgrad_generic": (lambda a: 2.0 * a[0] * a[4] * a[5] * a[3] * a[6] * a[7] * a[8], "direct"),
    "smsut_conv2d_wgrad_generic": (lambda a: 2.0 * a[0] * a[4] * a[5] * a[3] * a[6] * a[7] * a[8], "direct"),
    "smsut_conv2d_fwd_mfma": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * _k2(a, 5), "mfma"),
    # two image sets of one layer in one launch (ints: NA, NB, ca, H, W, Cin, Cout); conv_flops_of counts the fused-shortcut row too
    "smsut_conv2d_wgrad_pair": (lambda a: 2.0 * (a[0] + a[1]) * a[3] * a[4] * a[5] * a[6] * 9, "mfma"),
    "smsut_conv2d_wgrad_pair_slabs": (lambda a: 2.0 * (a[0] + a[1]) * a[2] * a[3] * a[4] * a[5] * 9, "mfma"),
    "smsut_conv2d_fwd_mfma_stats": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * _k2(a, 5), "mfma"),
    "smsut_conv2d_dgrad_mfma_bwdstats": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_mfma": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * _k2(a, 5), "mfma"),
    "smsut_conv2d_fwd_mfma_stats_inaff": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_mfma_inaff": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_mfma_slabs": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),   # (bench's roofline leg only)
    "smsut_conv2d_fwd_mfma_stats_cat": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_mfma_cat": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * _k2(a, 6), "mfma"),
    "smsut_conv2d_fwd_mfma_split": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * 9, "mfma"),
    # conv1 + 1x1 shortcut fused (3x3 taps + the centre-tap shortcut = 10 "taps"): forward, data- and weight-gradient
    "smsut_conv2d_fwd_mfma_stats_sc": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 10, "mfma"),
    "smsut_conv2d_dgrad_mfma_sc": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * 10, "mfma"),
    "smsut_conv2d_wgrad_mfma_sc": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * 10, "mfma"),
    "smsut_conv2d_k4_fwd": (lambda a: 32.0 * a[0] * (a[1] - 1) * (a[2] - 1) * a[3] * a[4], "mfma"),
    "smsut_conv2d_k4_wgrad": (lambda a: 32.0 * a[0] * (a[1] - 1) * (a[2] - 1) * a[3] * a[4], "mfma"),
    # fp16-operand forms (config 5)
    "smsut_conv2d_fwd_mfma_f16": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * _k2(a, 5), "mfma"),
    "smsut_conv2d_fwd_mfma_stats_f16": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * _k2(a, 5), "mfma"),
    "smsut_conv2d_fwd_mfma_stats_cat_f16": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_fwd_mfma_stats_sc_f16": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 10, "mfma"),
    "smsut_conv2d_fwd_mfma_split_f16": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * 9, "mfma"),
    "smsut_conv2d_dgrad_mfma_bwdstats_f16": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_f16": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * 9, "mfma"),
    "smsut_conv2d_dgrad_mfma_sc_f16": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * 10, "mfma"),
    # ... with fp16 storage of the block-internal raw conv outputs ("_hs")
    "smsut_conv2d_fwd_mfma_stats_f16_hs": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_fwd_mfma_stats_sc_f16_hs": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 10, "mfma"),
    "smsut_conv2d_fwd_mfma_stats_f16_hsx": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_f16_xh": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_f16_xh_inaff": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_fwd_mfma_stats_inaff_f16_hsx": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_dgrad_mfma_bwdstats_f16_hs": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3] * a[4] * 9, "mfma"),
    "smsut_conv2d_wgrad_sc_f16": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4] * a[5] * 10, "mfma"),
    "smsut_conv1x1_fwd": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3], "mfma"),
    "smsut_conv1x1_wgrad": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3], "mfma"),
    "smsut_conv1x1_fwd_cat": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4], "mfma"),
    "smsut_conv1x1_wgrad_cat": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4], "mfma"),
    "smsut_conv1x1_fwd_split": (lambda a: 2.0 * a[1] * a[2] * a[3] * a[4], "mfma"),
    "smsut_conv1x1_thin_dgrad": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3], "direct"),
    "smsut_conv1x1_thin_wgrad": (lambda a: 2.0 * a[0] * a[1] * a[2] * a[3], "direct"),
    "smsut_conv2d_small_fwd": (lambda a: 2.0 * a[0] * a[4] * a[5] * a[3] * a[6] * _k2(a, 7), "direct"),
    "smsut_conv2d_small_dgrad": (lambda a: 2.0 * a[0] * a[4] * a[5] * a[3] * a[6] * _k2(a, 7), "direct"),
    "smsut_conv2d_flat_wgrad": (lambda a: 2.0 * a[0] * a[4] * a[5] * a[3] * a[6] * _k2(a, 7), "mfma"),
    "smsut_convT2x2_fwd_mfma": (lambda a: 8.0 * a[0] * a[1] * a[2] * a[3] * a[4], "mfma"),
    "smsut_convT2x2_dgrad_mfma": (lambda a: 8.0 * a[0] * a[1] * a[2] * a[3] * a[4], "mfma"),
    "smsut_convT2x2_wgrad_mfma": (lambda a: 8.0 * a[0] * a[1] * a[2] * a[3] * a[4], "mfma"),
    "smsut_convT2x2_fwd_ps": (lambda a: 8.0 * a[0] * a[1] * a[2] * a[3] * a[4], "mfma"),
    "smsut_convT2x2_wgrad_ps": (lambda a: 8.0 * a[0] * a[1] * a[2] * a[3] * a[4], "mfma"),
}


# 3x3 fp32 forward / data-gradient entry points whose arithmetic depends on the shape (direct or Winograd, smsut_conv2d_mfma_form):
# entry point -> (ints -> (N, H, W, Kdim, Ndim, sc_dgrad), executed / algorithmic products when a Winograd form runs).
# Winograd F(2x2,3x3): 16 instead of 36 products per 2x2 output tile and channel pair (1 / 2.25).  Fused shortcut FORWARD: the 1x1
# conv runs on the tile's raw pixels, 4 more products (20 of 40 algorithmic = 0.5); fused shortcut DATA-GRADIENT: the shortcut's
# gradient is a second reduction half that goes through the transform like the first (32 of 40 = 0.8).
_WINO = 16.0 / 36.0
_FORM: Dict[str, Tuple[Callable[[List[int]], Tuple[int, int, int, int, int, int]], float]] = {
    "smsut_conv2d_fwd_mfma": (lambda a: (a[0], a[1], a[2], a[3], a[4], 0) if a[5] == 3 else None, _WINO),
    "smsut_conv2d_fwd_mfma_stats": (lambda a: (a[0], a[1], a[2], a[3], a[4], 0) if a[5] == 3 else None, _WINO),
    "smsut_conv2d_dgrad_mfma_bwdstats": (lambda a: (a[0], a[1], a[2], a[3], a[4], 0), _WINO),
    "smsut_conv2d_fwd_mfma_stats_inaff": (lambda a: (a[0], a[1], a[2], a[3], a[4], 0), _WINO),
    "smsut_conv2d_fwd_mfma_stats_cat": (lambda a: (a[0], a[1], a[2], a[3], a[4], 0), _WINO),
    "smsut_conv2d_fwd_mfma_split": (lambda a: (a[1], a[2], a[3], a[4], a[5], 0) if a[6] == 3 else None, _WINO),
    "smsut_conv2d_fwd_mfma_stats_sc": (lambda a: (a[0], a[1], a[2], a[3], a[4], 0), 0.5),
    "smsut_conv2d_dgrad_mfma_sc": (lambda a: (a[1], a[2], a[3], 2 * a[4], a[5], 1), 0.8),
}


def _base(name: str) -> str:
    """`_pre` / `_fin` entry points (same call + the caller's prepared Winograd image / + the in-launch finalize) and `_amax` ones (same call + the output's absolute
    maximum handed to the fp16-operand consumer) count as the entry point they extend."""
    if name.endswith("_pre"):
        return name[:-4]
    if name.endswith("_fin"):                # (r05) the same launch, statistics finalised inside it: same integer arguments
        return name[:-4]
    return name[:-5] if name.endswith("_amax") else name



def _ints(name: str, conv_args) -> List[int]:
    sig = H.SIGNATURES[name].replace(" ", "")
    return [a for c, a in zip(sig, conv_args) if c in "il"]


def executed_factor(name: str, conv_args) -> float:
    """Share of a call's algorithmic conv FLOPs that the matrix pipes execute (1.0 unless a Winograd form takes the shape)."""
    ent = _FORM.get(_base(name))
    if ent is None:
        return 1.0
    q = ent[0](_ints(name, conv_args))
    if q is None or H.call("smsut_conv2d_mfma_form", *q) == 0:
        return 1.0
    return ent[1]


def _pair_dims(conv_args):
    """smsut_conv2d_wgrad_pair (two image sets of one layer in one launch): (images, H, W, Cin, Cout, tap rows)."""
    na, nb, _ca, h, w, cin, cout = [a for c, a in zip(H.SIGNATURES["smsut_conv2d_wgrad_pair"].replace(" ", ""), conv_args) if c == "i"]
    return na + nb, h, w, cin, cout, (10 if conv_args[3] is not None else 9)          # (conv_args[3] = gsA: fused shortcut row)


def conv_flops_of(name: str, conv_args) -> float:
    """Algorithmic FLOPs of one recorded call (0 for non-conv entry points)."""
    if name == "smsut_conv2d_wgrad_pair":
        n, h, w, cin, cout, rows = _pair_dims(conv_args)
        return 2.0 * n * h * w * cin * cout * rows
    ent = _CONV_FLOPS.get(_base(name))
    if ent is None:
        return 0.0
    sig = H.SIGNATURES[name].replace(" ", "")
    ints = [a for c, a in zip(sig, conv_args) if c in "il"]
    return float(ent[0](ints))


# ---- algorithmic bytes: every tensor a call must at least READ ONCE and WRITE ONCE (fp32), the rule SURVEY 8d derives its
# 0.62 GB-per-slice lower bound of the U-Net step from (conv inputs + conv outputs + InstanceNorm element passes), applied call by
# call to what THIS build launches -- so a class's measured HBM traffic (profiles/*_step_*_classes.md) has a floor to stand against.
# entry point (``_pre`` stripped) -> lambda(ints) -> bytes.  Calls that are not listed count 0 (the sum stays a lower bound).
def _cv(n, h, w, cin, cout, extra_out=0, extra_in=0):
    return 4.0 * n * h * w * (cin + extra_in + cout + extra_out)


_BYTES: Dict[str, Callable[[List[int]], float]] = {
    "smsut_conv2d_fwd_mfma": lambda a: _cv(a[0], a[1], a[2], a[3], a[4]),
    "smsut_conv2d_fwd_mfma_stats": lambda a: _cv(a[0], a[1], a[2], a[3], a[4]),
    "smsut_conv2d_dgrad_mfma_bwdstats": lambda a: _cv(a[0], a[1], a[2], a[3], a[4], extra_in=a[4]),     # + y1 (the mask)
    "smsut_conv2d_fwd_mfma_stats_inaff": lambda a: _cv(a[0], a[1], a[2], a[3], a[4]),
    "smsut_conv2d_fwd_mfma_stats_cat": lambda a: _cv(a[0], a[1], a[2], a[3], a[4]),
    "smsut_conv2d_fwd_mfma_split": lambda a: _cv(a[1], a[2], a[3], a[4], a[5]),
    "smsut_conv2d_fwd_mfma_stats_sc": lambda a: _cv(a[0], a[1], a[2], a[3], a[4], extra_out=a[4]),
    "smsut_conv2d_fwd_mfma_stats_sc_f16": lambda a: _cv(a[0], a[1], a[2], a[3], a[4], extra_out=a[4]),
    "smsut_conv2d_dgrad_mfma_sc": lambda a: _cv(a[1], a[2], a[3], 2 * a[4], a[5]),
    "smsut_conv2d_dgrad_mfma_sc_f16": lambda a: _cv(a[1], a[2], a[3], 2 * a[4], a[5]),
    "smsut_conv2d_wgrad_sc_f16": lambda a: _cv(a[1], a[2], a[3], a[4], 2 * a[5]),
    "smsut_conv2d_wgrad_mfma": lambda a: _cv(a[0], a[1], a[2], a[3], a[4]),
    "smsut_conv2d_wgrad_mfma_inaff": lambda a: _cv(a[0], a[1], a[2], a[3], a[4]),
    "smsut_conv2d_wgrad_mfma_cat": lambda a: _cv(a[1], a[2], a[3], a[4], a[5]),
    "smsut_conv2d_wgrad_mfma_sc": lambda a: _cv(a[1], a[2], a[3], a[4], 2 * a[5]),
    "smsut_conv1x1_fwd": lambda a: 4.0 * a[0] * a[1] * (a[2] + a[3]),
    "smsut_conv1x1_wgrad": lambda a: 4.0 * a[0] * a[1] * (a[2] + a[3]),
    "smsut_convT2x2_fwd_mfma": lambda a: 4.0 * a[0] * a[1] * a[2] * (a[3] + 4 * a[4]),
    "smsut_convT2x2_dgrad_mfma": lambda a: 4.0 * a[0] * a[1] * a[2] * (a[3] + 4 * a[4]),
    "smsut_convT2x2_wgrad_mfma": lambda a: 4.0 * a[0] * a[1] * a[2] * (a[3] + 4 * a[4]),
    "smsut_convT2x2_fwd_ps": lambda a: 4.0 * a[0] * a[1] * a[2] * (a[3] + 4 * a[4]),
    "smsut_convT2x2_wgrad_ps": lambda a: 4.0 * a[0] * a[1] * a[2] * (a[3] + 4 * a[4]),
    # InstanceNorm / residual tails: (n, hw, c) -> tensor passes of n * hw * c floats
    "smsut_restail_fwd": lambda a: 4.0 * a[0] * a[1] * a[2] * 3,            # y2, s (or x) -> out
    "smsut_restail_bwd": lambda a: 4.0 * a[0] * a[1] * a[2] * 5,            # g, y2, s -> gy2, gs (one ideal pass; the kernel needs two)
    "smsut_in_apply_bwd": lambda a: 4.0 * a[0] * a[1] * a[2] * 3,           # gz, y1 -> gy1
    "smsut_instnorm_fwd": lambda a: 4.0 * a[0] * a[1] * a[2] * 2,
    "smsut_instnorm_fwd_partials": lambda a: 4.0 * a[1] * a[2] * a[3] * 2,
    "smsut_instnorm_bwd": lambda a: 4.0 * a[0] * a[1] * a[2] * 3,
    "smsut_instnorm_bwd2": lambda a: 4.0 * a[0] * a[1] * a[2] * 5,
    # pooling folded into the neighbouring node (r05): (n, h, w, c[, hs]); idx = one byte per pooled element
    "smsut_restail_fwd_pool": lambda a: a[0] * a[1] * a[2] * a[3] * ((2.0 * 2 if a[4] else 4.0 * 2) + 4.0 * 1.25 + 0.25),
    "smsut_restail_bwd_pool": lambda a: a[0] * a[1] * a[2] * a[3] * ((2.0 * 2 if a[4] else 4.0 * 2) + 4.0 * 3.25 + 0.25),
    "smsut_instnorm_pool_fwd_partials": lambda a: 4.0 * a[1] * a[2] * a[3] * a[4] * 1.25,      # x -> pooled
    "smsut_instnorm_pool_bwd": lambda a: 4.0 * a[0] * a[1] * a[2] * a[3] * 2.25,               # gp, x -> gx
    "smsut_maxpool2_fwd": lambda a: 4.0 * a[0] * a[1] * a[2] * a[3] * 1.25,
    "smsut_maxpool2_bwd": lambda a: 4.0 * a[0] * a[1] * a[2] * a[3] * 2.25,
    "smsut_maxpool2_bwd_add": lambda a: 4.0 * a[0] * a[1] * a[2] * a[3] * 3.25,
    "smsut_avgpool2_fwd": lambda a: 4.0 * a[0] * a[1] * a[2] * a[3] * 1.25,
    "smsut_avgpool2_bwd": lambda a: 4.0 * a[0] * a[1] * a[2] * a[3] * 1.25,
    "smsut_add_act": lambda a: 4.0 * a[0] * 3,
    # half storage (config 5): the fp16 tensors count 2 bytes per element
    "smsut_conv2d_fwd_mfma_stats_f16_hs": lambda a: a[0] * a[1] * a[2] * (4.0 * a[3] + 2.0 * a[4]),
    "smsut_conv2d_fwd_mfma_stats_sc_f16_hs": lambda a: a[0] * a[1] * a[2] * (4.0 * a[3] + 2.0 * 2 * a[4]),
    "smsut_conv2d_dgrad_mfma_bwdstats_f16_hs": lambda a: a[0] * a[1] * a[2] * (4.0 * a[3] + 4.0 * a[4] + 2.0 * a[4]),
    "smsut_restail_fwd_hs": lambda a: a[0] * a[1] * a[2] * (2.0 * 2 + 4.0),
    "smsut_restail_bwd_hs": lambda a: a[0] * a[1] * a[2] * (4.0 * 3 + 2.0 * 2),
    "smsut_in_apply_bwd_hs": lambda a: a[0] * a[1] * a[2] * (4.0 * 2 + 2.0),
    "smsut_instnorm_fwd_partials_hs": lambda a: a[1] * a[2] * a[3] * (2.0 + 4.0),
    "smsut_instnorm_fwd_partials_hs2": lambda a: a[1] * a[2] * a[3] * (2.0 + 2.0),
    "smsut_conv2d_fwd_mfma_stats_f16_hsx": lambda a: a[0] * a[1] * a[2] * (2.0 * a[3] + 2.0 * a[4]),
    "smsut_conv2d_wgrad_f16_xh": lambda a: a[0] * a[1] * a[2] * (2.0 * a[3] + 4.0 * a[4]),
    "smsut_conv2d_wgrad_f16_xh_inaff": lambda a: a[0] * a[1] * a[2] * (2.0 * a[3] + 4.0 * a[4]),
    "smsut_conv2d_fwd_mfma_stats_inaff_f16_hsx": lambda a: a[0] * a[1] * a[2] * (2.0 * a[3] + 2.0 * a[4]),
    "smsut_act_bwd": lambda a: 4.0 * a[0] * 3,
    "smsut_tanh_fwd": lambda a: 4.0 * a[0] * 2,
    "smsut_tanh_bwd": lambda a: 4.0 * a[0] * 3,
}


def bytes_of(name: str, conv_args) -> float:
    if name == "smsut_conv2d_wgrad_pair":
        n, h, w, cin, cout, rows = _pair_dims(conv_args)
        return _cv(n, h, w, cin, cout * (2 if rows == 10 else 1))
    ent = _BYTES.get(_base(name))
    return float(ent(_ints(name, conv_args))) if ent else 0.0


def _byte_class(name: str) -> str:
    n = _base(name)
    if "conv" in n:
        return "convolutions"
    if "restail" in n:
        return "residual tails"
    if "instnorm" in n or "in_apply" in n:
        return "InstanceNorm"
    return "pooling / pointwise"


class Row(collections.namedtuple("Row", "name args calls us flops exec_flops abytes")):
    """One distinct call of the step: ``flops`` = ALGORITHMIC conv FLOPs per call (2 N H W Cin Cout k^2), ``exec_flops``
    = what the matrix pipes execute for it (smaller where a Winograd form runs)."""
    @property
    def total_us(self):
        return self.us * self.calls


# Calls that must not be REPLAYED from a recording: their pointer arguments name device-side TABLES of further pointers, which are only
# valid at the moment of the call (the one-launch SGD step: baseTrainer.SgdStepper rebuilds its table when a gradient moves, and a replay
# of the old call would follow dangling entries -- it aborted the byte census of bench.py the first time the two met).
_NO_REPLAY = {"smsut_sgd_momentum_multi"}


def record_step(step: Callable[[], object]):
    """Run ``step()`` once with every status-returning C-ABI call recorded: [(entry point, converted args)]."""
    rec = []
    orig = H.call

    def spy(name, *args):
        if name not in H._NO_STATUS and name not in _NO_REPLAY:
            rec.append((name, [H.ptr(a) if isinstance(a, torch.Tensor) or a is None else a for a in args]))
        return orig(name, *args)
    H.call = spy
    try:
        step()
        torch.cuda.synchronize()
    finally:
        H.call = orig
    return rec



def replay(rec, reps: int = 8) -> List[Row]:
    """Replay each distinct (entry point, non-pointer arguments) ``reps`` times between HIP events."""
    lib = H.load()
    groups = collections.OrderedDict()
    for name, conv in rec:
        sig = H.SIGNATURES[name].replace(" ", "")
        key = (name, tuple(a for c, a in zip(sig, conv) if c in "ilfd"))
        groups.setdefault(key, []).append(conv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    # (`_pre` calls carry their prepared Winograd image as an argument: the replay runs the form the step ran; the image memory
    #  belongs to the module's ``_smsut_wino_set`` and may be one optimizer step old -- timing only)
    rows = []
    for (name, shp), calls in groups.items():
        conv = list(calls[0])
        fn = getattr(lib, name)
        conv[-1] = st.cuda_stream                        # the recorded stream may have been a capture stream
        for _ in range(2):
            fn(*conv)
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(reps):
            fn(*conv)
        e1.record(st)
        torch.cuda.synchronize()
        fl = conv_flops_of(name, calls[0])
        rows.append(Row(name, shp, len(calls), e0.elapsed_time(e1) / reps * 1e3, fl, fl * executed_factor(name, calls[0]),
                        bytes_of(name, calls[0])))
    return rows


def summarize(rows: List[Row], peak_tflops: float) -> dict:
    """Conv-side roofline of the step and the per-family split.  ``*_algorithmic`` / ``tflops`` count the conv FLOPs the layer
    DEFINES (what the contract's ``achieved`` is made of); ``*_executed`` count the products the MFMA pipes actually run (Winograd
    forms: 16 / 36 of them) -- the figure to hold against the 157.3 TFLOP/s the pipes can do."""
    conv = [r for r in rows if r.flops > 0]
    mfma = [r for r in conv if _CONV_FLOPS[_base(r.name)][1] == "mfma"]
    t_all = sum(r.total_us for r in rows)
    t_conv = sum(r.total_us for r in conv)
    f_conv = sum(r.flops * r.calls for r in conv)
    x_conv = sum(r.exec_flops * r.calls for r in conv)
    t_mfma = sum(r.total_us for r in mfma)
    f_mfma = sum(r.flops * r.calls for r in mfma)
    x_mfma = sum(r.exec_flops * r.calls for r in mfma)
    fam = collections.OrderedDict()
    for r in conv:
        k = ("wgrad" if "wgrad" in r.name else "fwd/dgrad") + (" 1x1" if "1x1" in r.name else (" convT" if "convT" in r.name else ""))
        t, f, x = fam.get(k, (0.0, 0.0, 0.0))
        fam[k] = (t + r.total_us, f + r.flops * r.calls, x + r.exec_flops * r.calls)
    byc = collections.OrderedDict()
    for r in rows:
        if r.abytes > 0:
            k = _byte_class(r.name)
            t, b = byc.get(k, (0.0, 0.0))
            byc[k] = (t + r.total_us, b + r.abytes * r.calls)
    tf = lambda f, t: round(f / (t * 1e-6) / 1e12, 2) if t else None                      # noqa: E731
    fr = lambda f, t: round(f / (t * 1e-6) / 1e12 / peak_tflops, 4) if t else None        # noqa: E731
    return {
        "kernel_ms_all": round(t_all / 1e3, 3), "kernel_ms_conv": round(t_conv / 1e3, 3),
        "conv_gflop": round(f_conv / 1e9, 2), "conv_gflop_mfma_executed": round(x_conv / 1e9, 2),
        "step_conv_tflops": tf(f_conv, t_conv), "step_conv_frac": fr(f_conv, t_conv),
        "step_conv_tflops_mfma_executed": tf(x_conv, t_conv), "step_conv_frac_mfma_executed": fr(x_conv, t_conv),
        "mfma_conv_frac_algorithmic": fr(f_mfma, t_mfma), "mfma_conv_frac_executed": fr(x_mfma, t_mfma),
        "conv_share_of_kernel_time": round(t_conv / t_all, 4) if t_all else None,
        # every tensor of a call read once + written once (fp32): the floor of this build's launches, per class (GB per step) with
        # the rate these calls reach against it (GB/s of ALGORITHMIC bytes; HBM peak 8000)
        "algorithmic_gbytes": round(sum(b for _, b in byc.values()) / 1e9, 3),
        "algorithmic_bytes_by_class": {k: {"gbytes": round(b / 1e9, 3), "ms": round(t / 1e3, 3), "gbs": round(b / (t * 1e-6) / 1e9, 1)}
                                       for k, (t, b) in byc.items()},
        "families": {k: {"ms": round(t / 1e3, 3), "algorithmic_tflops": round(f / (t * 1e-6) / 1e12, 1),
                         "mfma_executed_tflops": round(x / (t * 1e-6) / 1e12, 1)} for k, (t, f, x) in fam.items()},
    }


def table(rows: List[Row], min_share: float = 0.007) -> str:
    tot = sum(r.total_us for r in rows) or 1.0
    out = []
    for r in sorted(rows, key=lambda r: -r.total_us):
        if r.total_us / tot < min_share:
            break
        tf = f"{r.flops / (r.us * 1e-6) / 1e12:6.1f} TF" if r.flops else "         "
        out.append(f"{r.total_us / 1e3:7.3f} ms {100 * r.total_us / tot:5.1f}%  {r.calls:3d} x {r.us:7.1f} us {tf}  "
                   f"{r.name.replace('smsut_', '')} {r.args}")
    return "\n".join(out)


def lost_table(rows: List[Row], peak_tflops: float) -> str:
    """Every row, sorted by the time it would give back at the fp32 MFMA peak (conv rows) -- where the step's gap to the
    roofline sits, shape by shape; non-conv rows (HBM-bound passes) are listed with their whole time."""
    tot = sum(r.total_us for r in rows) or 1.0
    ent = []
    for r in rows:
        ideal = r.flops / (peak_tflops * 1e12) * 1e6 if r.flops else 0.0
        ent.append((r.total_us - ideal * r.calls, r))
    out = []
    for lost, r in sorted(ent, key=lambda e: -e[0]):
        tf = f"{r.flops / (r.us * 1e-6) / 1e12:6.1f} TF" if r.flops else "         "
        out.append(f"lost {lost / 1e3:7.3f} ms  total {r.total_us / 1e3:7.3f} ms {100 * r.total_us / tot:5.1f}%  {r.calls:3d} x {r.us:7.1f} us {tf}  "
                   f"{r.name.replace('smsut_', '')} {r.args}")
    return "\n".join(out)
