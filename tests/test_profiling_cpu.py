"""The per-shape profiler's FLOP table (smsut_amd.profiling): every conv-like entry point of the C ABI that the trainers call
must have an algorithmic-FLOP formula, and the formulas must agree with SURVEY.md 8d's count (2 * N * Ho * Wo * Cin * Cout * taps)
-- otherwise roofline.step_conv_frac silently drops kernels from the conv side."""
import re

import pytest

from smsut_amd import _hip as H
from smsut_amd import profiling


def _args(name, **ints):
    """Argument list in signature order: pointers / stream 0, floats 0.01, ints from ``ints`` by position."""
    sig = H.SIGNATURES[name].replace(" ", "")
    vals = iter(ints.values())
    return [next(vals) if c in "il" else (0.01 if c in "fd" else 0) for c in sig]


def test_conv_entry_points_have_flop_formulas():
    conv_like = [n for n in H.SIGNATURES if re.search(r"conv(2d|1x1|T2x2)_.*(fwd|dgrad|wgrad)", n)
                 and not re.search(r"_ws$|supported|tiles|persistent|_cfg$", n)]          # (_cfg: tuning hook, never called by a trainer)
    # (`_pre` entry points = the same call + the caller's prepared Winograd image: they count as the entry point they extend)
    missing = [n for n in conv_like if profiling._base(n) not in profiling._CONV_FLOPS]
    assert not missing, missing


def test_executed_flops_and_algorithmic_bytes_tables_cover_the_pre_forms():
    """``_pre`` calls are accounted like their base entry point (FLOPs, executed factor table, algorithmic bytes)."""
    a = _args("smsut_conv2d_fwd_mfma_stats_pre", N=4, H=64, W=64, K=64, Nd=64, KS=3)
    b = _args("smsut_conv2d_fwd_mfma_stats", N=4, H=64, W=64, K=64, Nd=64, KS=3)
    assert profiling.conv_flops_of("smsut_conv2d_fwd_mfma_stats_pre", a) == profiling.conv_flops_of("smsut_conv2d_fwd_mfma_stats", b) > 0
    assert profiling.bytes_of("smsut_conv2d_fwd_mfma_stats_pre", a) == 4.0 * 4 * 64 * 64 * (64 + 64)
    assert profiling.bytes_of("smsut_restail_bwd", _args("smsut_restail_bwd", n=16, hw=65536, c=16)) == 4.0 * 16 * 65536 * 16 * 5
    assert profiling._base("smsut_conv2d_fwd_mfma_stats_sc_pre") in profiling._FORM


@pytest.mark.parametrize("name,ints,taps", [
    ("smsut_conv2d_fwd_mfma_stats", dict(N=4, H=64, W=64, K=16, Nd=32, KS=3), 9),
    ("smsut_conv2d_fwd_mfma_stats_sc", dict(N=4, H=64, W=64, K=16, Nd=32), 10),
    ("smsut_conv2d_dgrad_mfma_sc", dict(split=0, N=4, H=64, W=64, Co=16, Ci=32), 10),
    ("smsut_conv2d_wgrad_mfma_sc", dict(ca=0, N=4, H=64, W=64, Ci=32, Co=64), 10),
    ("smsut_conv2d_wgrad_mfma", dict(N=4, H=64, W=64, Ci=32, Co=64, KS=3), 9),
    ("smsut_conv1x1_fwd", dict(N=4, HW=4096, K=16, Nd=32, tr=0), 1),
])
def test_flop_formulas(name, ints, taps):
    v = list(ints.values())
    n = ints["N"]
    hw = ints["HW"] if "HW" in ints else ints["H"] * ints["W"]
    chans = [x for k, x in ints.items() if k in ("K", "Nd", "Co", "Ci")]
    want = 2.0 * n * hw * chans[0] * chans[1] * taps
    assert profiling.conv_flops_of(name, _args(name, **ints)) == want, (name, v)
