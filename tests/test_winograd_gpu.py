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


class _Prepared:
    """smsut_wino_prepare for one weight tensor, both forms; the images are kept alive by the object and PASSED to the `_pre` entry
    points by ``_call`` (r04: the library keeps no table of images -- an image is an argument)."""

    def __init__(self, H, w, ci, co):
        import ctypes
        forms = [(0, ci, co), (1, co, ci)]
        self.u = [torch.full((max(H.call("smsut_wino_image_floats", k, m), 1),), float("nan"), device="cuda") for _, k, m in forms]
        PA, IA = ctypes.c_void_p * 2, ctypes.c_int * 2
        self.arr = (PA(w.data_ptr(), w.data_ptr()), PA(*[u.data_ptr() for u in self.u]), IA(ci, co), IA(co, ci), IA(0, 1))
        addr = [ctypes.addressof(a) for a in self.arr]
        H.call("smsut_wino_prepare", *addr, 2, H.stream_ptr())


def _call(H, keep, name, tr, *args):
    """the entry point, or -- with prepared images -- its `_pre` form with the image of this form (tr: 0 forward, 1 data-gradient)
    as the argument in front of the stream"""
    if keep is None:
        return H.call(name, *args)
    return H.call(name + "_pre", *args[:-1], keep.u[tr & 1], args[-1])


@pytest.fixture(params=[False, True], ids=["on-the-fly", "prepared"])
def prepared(request):
    """every form test runs twice: weights transformed inside the kernel, and copied from the caller's prepared image"""
    return request.param


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("n,h,ci,co", SHAPES)
def test_forward_datagrad_accumulate_statistics(n, h, ci, co, prepared):
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    g, x, w = _mk(n, h, ci, co, 3)
    keep = _Prepared(H, w, ci, co) if prepared else None  # noqa: F841
    ref = _conv64(x, w)
    y = torch.full((n, h, h, co), float("nan"), device="cuda")
    _call(H, keep, "smsut_conv2d_fwd_mfma", 0, x, w, y, n, h, h, ci, co, 3, 0, st)
    assert _rel(y, ref) < 2e-6
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, 0)
    ys, part = torch.full_like(y, float("nan")), torch.full((n * tiles * co * 2,), float("nan"), device="cuda")
    _call(H, keep, "smsut_conv2d_fwd_mfma_stats", 0, x, w, ys, part, n, h, h, ci, co, 3, st)
    assert torch.equal(ys, y)                                     # same kernel, same order
    p = part.view(n, tiles, co, 2).double().sum(1)
    assert torch.allclose(p[..., 0], ys.double().sum((1, 2)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(p[..., 1], (ys.double() ** 2).sum((1, 2)), rtol=1e-5, atol=1e-3)
    gy = torch.randn(n, h, h, co, generator=g).cuda()
    refd = _dgrad64(gy, w)
    gx = torch.full((n, h, h, ci), float("nan"), device="cuda")
    _call(H, keep, "smsut_conv2d_fwd_mfma", 1, gy, w, gx, n, h, h, co, ci, 3, 1, st)
    assert _rel(gx, refd) < 2e-6
    base = torch.randn(n, h, h, ci, generator=g).cuda()
    acc = base.clone()
    _call(H, keep, "smsut_conv2d_fwd_mfma", 1, gy, w, acc, n, h, h, co, ci, 3, 3, st)                 # accumulate form
    assert _rel(acc, refd + base.double()) < 2e-6


@pytest.mark.parametrize("n,h,c", [(3, 32, 32), (3, 32, 64), (2, 16, 128), (2, 16, 256), (5, 32, 96)])
def test_input_side_instnorm_and_bst_forms(n, h, c, prepared):
    """conv2 of a BasicBlock on the raw conv1 output (InstanceNorm + LeakyReLU while staging; zero padding applies AFTER the
    affine) and its data-gradient with the LeakyReLU mask / InstanceNorm-backward partials in the epilogue (blocks.py:66-72)."""
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    g, y1, w = _mk(n, h, c, c, 5)
    keep = _Prepared(H, w, c, c) if prepared else None  # noqa: F841
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
    _call(H, keep, "smsut_conv2d_fwd_mfma_stats_inaff", 0, y1, w, y2, part, mean.float().contiguous(), rstd.float().contiguous(), gam, bet, slope,
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
    _call(H, keep, "smsut_conv2d_dgrad_mfma_bwdstats", 1, gy, w, gz, pb, y1, mean.float().contiguous(), rstd.float().contiguous(), gam, bet, slope,
           n, h, h, c, c, st)
    flips = (gz.double() - refz).abs() > 1e-4 * refz.abs().max()                    # (a pre-activation within rounding of 0 may flip)
    assert flips.float().mean() < 1e-4
    assert float(((gz.double() - refz).abs() * ~flips).max() / refz.abs().max()) < 5e-6
    xhat = (yd - mean[:, None, None]) * rstd[:, None, None]
    q = pb.view(n, tiles, c, 2).double().sum(1)
    assert torch.allclose(q[..., 0], gz.double().sum((1, 2)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(q[..., 1], (gz.double() * xhat).sum((1, 2)), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("n,h,ci,co", [(3, 32, 64, 32), (2, 32, 128, 64), (2, 16, 256, 128), (3, 32, 32, 16)])
def test_virtual_cat_split_and_fused_shortcut_forms(n, h, ci, co, prepared):
    """Decoder conv1 on cat([up, skip]) read in place, the split-output data-gradient, and conv1 + 1x1 shortcut fused
    (blocks.py:37-50, 66-80) against fp64."""
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    g, x, w = _mk(n, h, ci, co, 9)
    keep = _Prepared(H, w, ci, co) if prepared else None  # noqa: F841
    ca = ci // 2
    xa, xb = x[..., :ca].contiguous(), x[..., ca:].contiguous()
    ref = _conv64(x, w)
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, 0)
    y, part = torch.full((n, h, h, co), float("nan"), device="cuda"), torch.zeros(n * tiles * co * 2, device="cuda")
    _call(H, keep, "smsut_conv2d_fwd_mfma_stats_cat", 0, xa, xb, w, y, part, n, h, h, ci, co, st)
    assert _rel(y, ref) < 2e-6
    gy = torch.randn(n, h, h, co, generator=g).cuda()
    refd = _dgrad64(gy, w)
    ga, gb = torch.full((n, h, h, ca), float("nan"), device="cuda"), torch.full((n, h, h, ci - ca), float("nan"), device="cuda")
    assert H.call("smsut_conv2d_mfma_split_supported", n, h, h, co, ci, ca) == 1
    _call(H, keep, "smsut_conv2d_fwd_mfma_split", 1, gy, w, ga, gb, ca, n, h, h, co, ci, 1, st)
    assert _rel(torch.cat([ga, gb], 3), refd) < 2e-6
    if H.call("smsut_conv2d_fwd_sc_supported", n, h, h, ci, co, 1):
        w1 = (torch.randn(ci, co, generator=g) / np.sqrt(ci)).cuda()
        y2, s2 = torch.full_like(y, float("nan")), torch.full_like(y, float("nan"))
        p2, q2 = torch.zeros_like(part), torch.zeros_like(part)
        _call(H, keep, "smsut_conv2d_fwd_mfma_stats_sc", 0, xa, xb, w, w1, y2, s2, p2, q2, n, h, h, ci, co, st)
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


@pytest.mark.parametrize("n,h,ci,co", [(3, 32, 64, 64), (2, 16, 256, 128), (5, 16, 96, 32), (2, 32, 128, 48)])
def test_prepared_weights_are_bit_identical_and_explicit(n, h, ci, co):
    """A prepared image passed with the call gives the bits of the on-the-fly transform (same arithmetic, done once); the library
    remembers nothing between calls (r04: no binding table) -- the plain entry point right after a `_pre` call transforms on the fly
    again; a stale image is the caller's business -- shown here on purpose: the kernel really reads the image it is handed."""
    import smsut_amd  # noqa: F401
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    g, x, w = _mk(n, h, ci, co, 21)
    gy = torch.randn(n, h, h, co, generator=g).cuda()

    def run(keep):
        y, gx = torch.full((n, h, h, co), float("nan"), device="cuda"), torch.full((n, h, h, ci), float("nan"), device="cuda")
        _call(H, keep, "smsut_conv2d_fwd_mfma", 0, x, w, y, n, h, h, ci, co, 3, 0, st)
        _call(H, keep, "smsut_conv2d_fwd_mfma", 1, gy, w, gx, n, h, h, co, ci, 3, 1, st)
        return y, gx
    y0, g0 = run(None)
    keep = _Prepared(H, w, ci, co)
    y1, g1 = run(keep)
    assert torch.equal(y0, y1) and torch.equal(g0, g1)
    keep.u[0].mul_(2.0)                                    # the forward image no longer matches the weights ...
    y2, g2 = run(keep)
    assert torch.equal(y2, 2.0 * y0) and torch.equal(g2, g0)   # ... and it is the image that is read (x2 is exact in fp32)
    y3, g3 = run(None)                                     # nothing was remembered: the plain entry points read the weights
    assert torch.equal(y3, y0) and torch.equal(g3, g0)


def test_prepared_scope_in_a_training_step_is_bit_identical():
    """ops.wino_prepared (what the trainers wrap their phases in): the same forward + backward with and without the scope."""
    import smsut_amd  # noqa: F401
    from smsut_amd import ops
    from smsut_amd.network.blocks import BasicBlock
    torch.manual_seed(5)
    blk = BasicBlock(64, 128, norm="instance", act="lrelu").cuda()
    x = ops.nhwc(torch.randn(3, 64, 32, 32, device="cuda"))
    outs = []
    for scoped in (False, True, True):
        for p in blk.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        if scoped:
            with ops.wino_prepared(blk):
                y = blk(xi)
                y.square().mean().backward()
        else:
            y = blk(xi)
            y.square().mean().backward()
        outs.append([y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in blk.parameters()])
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)
    # weights updated the way the trainers do it (fused optimizer: in place, ``_version`` does not move): the next scope must
    # see the new weights
    opt = torch.optim.SGD(blk.parameters(), lr=0.5, fused=True)
    opt.step()
    res = []
    for scoped in (True, False):
        xi = x.clone()
        with torch.no_grad():
            if scoped:
                with ops.wino_prepared(blk, forms="f"):
                    res.append(blk(xi).clone())
            else:
                res.append(blk(xi).clone())
    assert torch.equal(res[0], res[1]) and not torch.equal(res[0], outs[0][0])
    ws = blk.__dict__["_smsut_wino_set"]
    assert ws.forms[0].n >= 1 and ws.forms[1].n >= 1
