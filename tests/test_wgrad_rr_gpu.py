"""Register-row 3x3 weight gradient (csrc/conv_wgrad_rr.hip, r04) through the C ABI -- reference op: d loss / d weight of conv3x3
inside BasicBlock (/root/reference/network/blocks.py:10-12, :53-80) -- against fp64 torch on a sweep that covers every variant of the
kernel (16x16 / 32x16 / 16x32 / 32x32 tiles per wave, ring of 8 and of 4 rows), every form (plain, virtual cat, input-side
InstanceNorm + LeakyReLU, fused 1x1 shortcut, both together), image borders on every side (strips at the left / right edge, units at
the top / bottom, planes of ONE strip, odd batch sizes, non-square planes) and the bit-level promises: fused-shortcut rows 0..8 ==
the plain call, virtual cat == materialised cat, slabs-only entry point + reduction == the full call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, gy, gs=None, aff=None):
    xd = x.double().permute(0, 3, 1, 2)
    if aff is not None:
        mean, rstd, gamma, beta, slope = aff
        v = (xd - mean.double()[:, :, None, None]) * (rstd.double()[:, :, None, None] * gamma.double()[None, :, None, None]) \
            + beta.double()[None, :, None, None]
        xd = torch.where(v > 0, v, v * slope)
    ci, co = x.shape[-1], gy.shape[-1]
    gw = torch.nn.grad.conv2d_weight(xd, (co, ci, 3, 3), gy.double().permute(0, 3, 1, 2), padding=1).permute(2, 3, 1, 0).reshape(9, ci, co)
    if gs is not None:
        g1 = torch.nn.grad.conv2d_weight(xd, (co, ci, 1, 1), gs.double().permute(0, 3, 1, 2)).permute(2, 3, 1, 0).reshape(1, ci, co)
        gw = torch.cat([gw, g1], 0)
    return gw


SHAPES = [  # n, h, w, ci, co
    (3, 64, 64, 16, 16), (1, 32, 16, 16, 16), (2, 20, 48, 16, 16),       # 16x16 per wave: ring of 8, a ONE-strip plane, ring of 4 (h % 8 != 0)
    (5, 40, 32, 32, 16), (2, 12, 16, 32, 16),                            # 32x16
    (3, 24, 80, 16, 32), (1, 8, 16, 16, 32),                             # 16x32
    (2, 32, 32, 32, 32), (3, 16, 48, 64, 32), (1, 4, 16, 32, 96), (4, 16, 16, 128, 64), (2, 8, 32, 256, 32),   # 32x32 per wave
]


@pytest.mark.parametrize("n,h,w,ci,co", SHAPES)
@pytest.mark.parametrize("form", ["plain", "cat", "inaff", "sc", "cat+sc"])
def test_register_row_weight_gradient_forms(n, h, w, ci, co, form):
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    if "cat" in form and ci < 32:
        pytest.skip("virtual cat needs two whole 16-channel halves")
    g = torch.Generator(device="cpu").manual_seed(n * 1000 + h * 10 + ci)
    x = torch.randn(n, h, w, ci, generator=g).cuda()
    gy = torch.randn(n, h, w, co, generator=g).cuda()
    gs = torch.randn(n, h, w, co, generator=g).cuda() if "sc" in form else None
    aff = None
    if form == "inaff":
        aff = ((torch.randn(n, ci, generator=g) * 0.3).cuda(), (torch.rand(n, ci, generator=g) + 0.5).cuda(),
               (torch.rand(ci, generator=g) + 0.5).cuda(), (torch.randn(ci, generator=g) * 0.2).cuda(), 0.01)
    want = _ref(x, gy, gs, aff)
    rows = 10 if gs is not None else 9
    gw = torch.full((rows, ci, co), float("nan"), device="cuda")
    ca = ci // 2
    xa, xb = (x[..., :ca].contiguous(), x[..., ca:].contiguous()) if "cat" in form else (x, None)
    if gs is not None:
        assert H.call("smsut_conv2d_wgrad_sc_supported", n, h, w, ci, co) == 1
        ws = torch.empty(H.call("smsut_conv2d_wgrad_sc_ws", n, h, w, ci, co), device="cuda")
        H.call("smsut_conv2d_wgrad_mfma_sc", xa, xb, ca if xb is not None else 0, gy, gs, gw, ws, n, h, w, ci, co, st)
    else:
        ws = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, w, ci, co, 3), device="cuda")
        if form == "inaff":
            m, r, ga, be, sl = aff
            H.call("smsut_conv2d_wgrad_mfma_inaff", x, gy, gw, ws, m, r, ga, be, sl, n, h, w, ci, co, st)
        elif form == "cat":
            H.call("smsut_conv2d_wgrad_mfma_cat", xa, xb, ca, gy, gw, ws, n, h, w, ci, co, 3, st)
        else:
            H.call("smsut_conv2d_wgrad_mfma", x, gy, gw, ws, n, h, w, ci, co, 3, st)
    assert torch.isfinite(gw).all()
    err = float((gw.double() - want).abs().max() / want.abs().max())
    assert err < 3e-6, (form, err)                                  # fp32 sums of n*h*w products against fp64
    # bit-level promises between the forms of one shape
    if form == "plain":
        # the slabs-only measurement entry point (bench.py's roofline leg) + the reduction == the full call
        ws2 = torch.empty_like(ws)
        slabs = H.call("smsut_conv2d_wgrad_mfma_slabs", x, gy, ws2, None, None, None, None, 0.0, n, h, w, ci, co, st)
        assert slabs > 0
        red = ws2[: slabs * 9 * ci * co].view(slabs, 9 * ci * co)
        acc = red[0].clone()
        for i in range(1, slabs):                                   # same fixed order as sum_splits
            acc += red[i]
        assert torch.allclose(acc, gw.view(-1), rtol=1e-5, atol=1e-5 * float(gw.abs().max()))
    if form in ("sc", "cat+sc"):
        g9 = torch.empty(9, ci, co, device="cuda")
        w9 = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, w, ci, co, 3), device="cuda")
        if xb is not None:
            H.call("smsut_conv2d_wgrad_mfma_cat", xa, xb, ca, gy, g9, w9, n, h, w, ci, co, 3, st)
        else:
            H.call("smsut_conv2d_wgrad_mfma", x, gy, g9, w9, n, h, w, ci, co, 3, st)
        assert torch.equal(gw[:9], g9)                              # the extra tile does not touch the 3x3 accumulators
    if form == "cat":
        g9 = torch.empty(9, ci, co, device="cuda")
        H.call("smsut_conv2d_wgrad_mfma", x, gy, g9, ws, n, h, w, ci, co, 3, st)
        assert torch.equal(gw, g9)                                  # virtual cat == materialised cat


PAIR_SHAPES = [  # na, nb, h, w, ci, co
    (3, 2, 64, 64, 16, 16), (1, 1, 32, 16, 16, 16), (2, 3, 40, 32, 32, 16), (2, 2, 24, 80, 16, 32), (1, 2, 32, 32, 32, 32),
    (3, 1, 16, 48, 64, 32), (2, 2, 16, 16, 128, 64), (8, 8, 32, 32, 128, 128),
]


@pytest.mark.parametrize("na,nb,h,w,ci,co", PAIR_SHAPES)
@pytest.mark.parametrize("form", ["plain", "cat", "inaff", "sc", "cat+sc"])
def test_paired_weight_gradient_is_the_sum_of_the_two_sets(na, nb, h, w, ci, co, form):
    """``smsut_conv2d_wgrad_pair``: ONE register-row launch over two image sets of the same layer (the two generator passes of a
    uganConsis iteration; the reference's autograd sums the two weight gradients, trainer/uganConsisTrainer.py:152,159,179) ==
    wgrad(A) + wgrad(B): against fp64 torch at the single-set bar, and against the two single-set calls."""
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    if "cat" in form and ci < 32:
        pytest.skip("virtual cat needs two whole 16-channel halves")
    cat, inaff, sc = "cat" in form, form == "inaff", "sc" in form
    if not H.call("smsut_conv2d_wgrad_pair_supported", na, nb, h, w, ci, co, int(cat), int(inaff), int(sc)):
        pytest.skip("shape not paired")
    g = torch.Generator(device="cpu").manual_seed(na * 1000 + nb * 100 + h + ci)
    ga, be, sl = (torch.rand(ci, generator=g) + 0.5).cuda(), (torch.randn(ci, generator=g) * 0.2).cuda(), 0.01
    sets = []
    for n in (na, nb):
        x = torch.randn(n, h, w, ci, generator=g).cuda()
        gy = torch.randn(n, h, w, co, generator=g).cuda()
        gs = torch.randn(n, h, w, co, generator=g).cuda() if sc else None
        m = (torch.randn(n, ci, generator=g) * 0.3).cuda() if inaff else None
        r = (torch.rand(n, ci, generator=g) + 0.5).cuda() if inaff else None
        sets.append((x, gy, gs, m, r))
    want = sum(_ref(x, gy, gs, (m, r, ga, be, sl) if inaff else None) for x, gy, gs, m, r in sets)
    rows = 10 if sc else 9
    ca = ci // 2
    parts = [((x[..., :ca].contiguous(), x[..., ca:].contiguous()) if cat else (x, None)) for x, *_ in sets]
    gw = torch.full((rows, ci, co), float("nan"), device="cuda")
    ws = torch.empty(H.call("smsut_conv2d_wgrad_pair_ws", na, nb, h, w, ci, co, int(cat), int(inaff), int(sc)), device="cuda")
    (xa0, xa1), (xb0, xb1) = parts
    H.call("smsut_conv2d_wgrad_pair", xa0, xa1, sets[0][1], sets[0][2], sets[0][3], sets[0][4], na,
           xb0, xb1, sets[1][1], sets[1][2], sets[1][3], sets[1][4], nb, ca if cat else 0,
           ga if inaff else None, be if inaff else None, sl, gw, ws, h, w, ci, co, st)
    assert torch.isfinite(gw).all()
    err = float((gw.double() - want).abs().max() / want.abs().max())
    assert err < 3e-6, (form, err)
    # ... and next to the two single-set calls (their sum differs from the paired accumulation by fp32 rounding only)
    single = torch.zeros(rows, ci, co, device="cuda")
    for (x, gy, gs, m, r), (p0, p1), n in zip(sets, parts, (na, nb)):
        o = torch.empty(rows, ci, co, device="cuda")
        if sc:
            w1 = torch.empty(H.call("smsut_conv2d_wgrad_sc_ws", n, h, w, ci, co), device="cuda")
            H.call("smsut_conv2d_wgrad_mfma_sc", p0, p1, ca if cat else 0, gy, gs, o, w1, n, h, w, ci, co, st)
        else:
            w1 = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, w, ci, co, 3), device="cuda")
            if inaff:
                H.call("smsut_conv2d_wgrad_mfma_inaff", x, gy, o, w1, m, r, ga, be, sl, n, h, w, ci, co, st)
            elif cat:
                H.call("smsut_conv2d_wgrad_mfma_cat", p0, p1, ca, gy, o, w1, n, h, w, ci, co, 3, st)
            else:
                H.call("smsut_conv2d_wgrad_mfma", x, gy, o, w1, n, h, w, ci, co, 3, st)
        single += o
    assert float((gw - single).abs().max() / single.abs().max()) < 2e-6


def test_conv_form_query_matches_the_dispatch():
    """``smsut_conv2d_mfma_form`` (what bench.py / profiling.py use to count the products the matrix pipes execute): 1 = resident
    Winograd for 16 / 32 reduction channels on planes divisible by 16, 2 = streamed-weight Winograd from 64, 0 = direct (8-channel
    forms, small grids, the fused shortcut data-gradient at 64 reduction channels)."""
    import os
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    if os.environ.get("SMSUT_WINOGRAD", "1") in ("0", ""):
        pytest.skip("Winograd forms switched off")
    assert H.call("smsut_conv2d_mfma_form", 16, 256, 256, 16, 16, 0) == 1
    assert H.call("smsut_conv2d_mfma_form", 16, 128, 128, 32, 32, 0) == 1
    assert H.call("smsut_conv2d_mfma_form", 16, 64, 64, 64, 64, 0) == 2
    assert H.call("smsut_conv2d_mfma_form", 16, 32, 32, 128, 128, 0) == 2
    assert H.call("smsut_conv2d_mfma_form", 16, 256, 256, 8, 16, 0) == 0
    assert H.call("smsut_conv2d_mfma_form", 16, 128, 128, 64, 32, 1) == 0        # fused shortcut data-gradient, 2 x 32 channels
    assert H.call("smsut_conv2d_mfma_form", 16, 4, 4, 256, 256, 0) == 0
