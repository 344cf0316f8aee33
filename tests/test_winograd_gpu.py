"""Winograd F(2x2, 3x3) kernels (r03) against fp64 torch, through the public C-ABI entry points.

Two kernel families take the fp32 3x3 stride-1 convs on planes divisible by 16: ``conv_mfma_fwd_p<..., WINO>`` (reductions of
16 / 32 channels, resident transformed weights) and ``conv_wino_l`` (csrc/conv_wino.hip: reductions >= 64 channels, weights
transformed on the fly, 1 or 2 output-channel slabs per wave).  tests/test_ops_gpu.py pins the fused forms of a shape to each
other bit for bit; here every form is pinned to an fp64 reference of reference network/blocks.py:10-12,53-80 arithmetic, on
shapes that reach both families, both slab counts (Ndim % 32 != 0 forces one), odd batch sizes and image borders."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(3, 32, 16, 16), (2, 48, 32, 16), (5, 32, 16, 32),                  # resident form
          (3, 32, 64, 64), (2, 48, 64, 48), (3, 16, 128, 64), (2, 32, 256, 128), (5, 16, 96, 32), (2, 64, 64, 32)]   # conv_wino_l


def _mk(n, h, ci, co, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, h, h, ci, generator=g).cuda()
    w = (torch.randn(3, 3, ci, co, generator=g) / np.sqrt(9 * ci)).cuda()          # [KH][KW][Cin][Cout] memory
    return g, x, w


def _conv64(x, w):            # x [n,h,w,ci], w [3,3,ci,co] -> [n,h,w,co], fp64
    return F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)


def _dgrad64(gy, w):          # gradient of conv w.r.t. its input
    return F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("n,h,ci,co", SHAPES)
def test_forward_datagrad_accumulate_statistics(n, h, ci, co):
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    g, x, w = _mk(n, h, ci, co, 3)
    ref = _conv64(x, w)
    y = torch.full((n, h, h, co), float("nan"), device="cuda")
    H.call("smsut_conv2d_fwd_mfma", x, w, y, n, h, h, ci, co, 3, 0, st)
    assert _rel(y, ref) < 2e-6
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, 0)
    ys, part = torch.full_like(y, float("nan")), torch.full((n * tiles * co * 2,), float("nan"), device="cuda")
    H.call("smsut_conv2d_fwd_mfma_stats", x, w, ys, part, n, h, h, ci, co, 3, st)
    assert torch.equal(ys, y)                                     # same kernel, same order
    p = part.view(n, tiles, co, 2).double().sum(1)
    assert torch.allclose(p[..., 0], ys.double().sum((1, 2)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(p[..., 1], (ys.double() ** 2).sum((1, 2)), rtol=1e-5, atol=1e-3)
    gy = torch.randn(n, h, h, co, generator=g).cuda()
    refd = _dgrad64(gy, w)
    gx = torch.full((n, h, h, ci), float("nan"), device="cuda")
    H.call("smsut_conv2d_fwd_mfma", gy, w, gx, n, h, h, co, ci, 3, 1, st)
    assert _rel(gx, refd) < 2e-6
    base = torch.randn(n, h, h, ci, generator=g).cuda()
    acc = base.clone()
    H.call("smsut_conv2d_fwd_mfma", gy, w, acc, n, h, h, co, ci, 3, 3, st)                 # accumulate form
    assert _rel(acc, refd + base.double()) < 2e-6


@pytest.mark.parametrize("n,h,c", [(3, 32, 32), (3, 32, 64), (2, 16, 128), (2, 16, 256), (5, 32, 96)])
def test_input_side_instnorm_and_bst_forms(n, h, c):
    """conv2 of a BasicBlock on the raw conv1 output (InstanceNorm + LeakyReLU while staging; zero padding applies AFTER the
    affine) and its data-gradient with the LeakyReLU mask / InstanceNorm-backward partials in the epilogue (blocks.py:66-72)."""
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    g, y1, w = _mk(n, h, c, c, 5)
    if not H.call("smsut_conv2d_mfma_persistent", n, h, h, c, c, 3, 0):
        pytest.skip("shape not on a fused-form kernel")
    slope, eps = 0.01, 1e-5
    gam, bet = (1 + 0.1 * torch.randn(c, generator=g)).cuda(), (0.1 * torch.randn(c, generator=g)).cuda()
    yd = y1.double()
    mean = yd.mean((1, 2)); var = yd.var((1, 2), unbiased=False); rstd = (var + eps).rsqrt()
    a1 = F.leaky_relu((yd - mean[:, None, None]) * rstd[:, None, None] * gam.double() + bet.double(), slope)
    ref = _conv64(a1, w)
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, c, c, 3, 0)
    y2, part = torch.full((n, h, h, c), float("nan"), device="cuda"), torch.zeros(n * tiles * c * 2, device="cuda")
    H.call("smsut_conv2d_fwd_mfma_stats_inaff", y1, w, y2, part, mean.float().contiguous(), rstd.float().contiguous(), gam, bet, slope,
           n, h, h, c, c, st)
    assert _rel(y2, ref) < 5e-6
    p = part.view(n, tiles, c, 2).double().sum(1)
    assert torch.allclose(p[..., 0], y2.double().sum((1, 2)), rtol=1e-5, atol=1e-3)
    # BST: gz = dgrad(gy) * lrelu'(affine(y1)); partials {sum gz, sum gz * xhat}
    gy = torch.randn(n, h, h, c, generator=g).cuda()
    pre = (yd - mean[:, None, None]) * rstd[:, None, None] * gam.double() + bet.double()
    mask = torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope))
    refz = _dgrad64(gy, w) * mask
    gz, pb = torch.full((n, h, h, c), float("nan"), device="cuda"), torch.zeros(n * tiles * c * 2, device="cuda")
    H.call("smsut_conv2d_dgrad_mfma_bwdstats", gy, w, gz, pb, y1, mean.float().contiguous(), rstd.float().contiguous(), gam, bet, slope,
           n, h, h, c, c, st)
    flips = (gz.double() - refz).abs() > 1e-4 * refz.abs().max()                    # (a pre-activation within rounding of 0 may flip)
    assert flips.float().mean() < 1e-4
    assert float(((gz.double() - refz).abs() * ~flips).max() / refz.abs().max()) < 5e-6
    xhat = (yd - mean[:, None, None]) * rstd[:, None, None]
    q = pb.view(n, tiles, c, 2).double().sum(1)
    assert torch.allclose(q[..., 0], gz.double().sum((1, 2)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(q[..., 1], (gz.double() * xhat).sum((1, 2)), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("n,h,ci,co", [(3, 32, 64, 32), (2, 32, 128, 64), (2, 16, 256, 128), (3, 32, 32, 16)])
def test_virtual_cat_split_and_fused_shortcut_forms(n, h, ci, co):
    """Decoder conv1 on cat([up, skip]) read in place, the split-output data-gradient, and conv1 + 1x1 shortcut fused
    (blocks.py:37-50, 66-80) against fp64."""
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    g, x, w = _mk(n, h, ci, co, 9)
    ca = ci // 2
    xa, xb = x[..., :ca].contiguous(), x[..., ca:].contiguous()
    ref = _conv64(x, w)
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, 0)
    y, part = torch.full((n, h, h, co), float("nan"), device="cuda"), torch.zeros(n * tiles * co * 2, device="cuda")
    H.call("smsut_conv2d_fwd_mfma_stats_cat", xa, xb, w, y, part, n, h, h, ci, co, st)
    assert _rel(y, ref) < 2e-6
    gy = torch.randn(n, h, h, co, generator=g).cuda()
    refd = _dgrad64(gy, w)
    ga, gb = torch.full((n, h, h, ca), float("nan"), device="cuda"), torch.full((n, h, h, ci - ca), float("nan"), device="cuda")
    assert H.call("smsut_conv2d_mfma_split_supported", n, h, h, co, ci, ca) == 1
    H.call("smsut_conv2d_fwd_mfma_split", gy, w, ga, gb, ca, n, h, h, co, ci, 1, st)
    assert _rel(torch.cat([ga, gb], 3), refd) < 2e-6
    if H.call("smsut_conv2d_fwd_sc_supported", n, h, h, ci, co, 1):
        w1 = (torch.randn(ci, co, generator=g) / np.sqrt(ci)).cuda()
        y2, s2 = torch.full_like(y, float("nan")), torch.full_like(y, float("nan"))
        p2, q2 = torch.zeros_like(part), torch.zeros_like(part)
        H.call("smsut_conv2d_fwd_mfma_stats_sc", xa, xb, w, w1, y2, s2, p2, q2, n, h, h, ci, co, st)
        assert torch.equal(y2, y)
        refs = (x.double().reshape(-1, ci) @ w1.double()).reshape(n, h, h, co)
        assert _rel(s2, refs) < 2e-6
        q = q2.view(n, tiles, co, 2).double().sum(1)
        assert torch.allclose(q[..., 1], (s2.double() ** 2).sum((1, 2)), rtol=1e-5, atol=1e-3)
    if H.call("smsut_conv2d_dgrad_sc_supported", n, h, h, co, ci, 0):
        w1 = (torch.randn(ci, co, generator=g) / np.sqrt(co)).cuda()
        gs = torch.randn(n, h, h, co, generator=g).cuda()
        got = torch.full((n, h, h, ci), float("nan"), device="cuda")
        H.call("smsut_conv2d_dgrad_mfma_sc", gy, gs, w, w1, got, None, 0, n, h, h, co, ci, st)
        assert _rel(got, refd + gs.double() @ w1.double().t()) < 2e-6
