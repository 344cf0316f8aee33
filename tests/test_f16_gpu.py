"""BASELINE config 5: the fp16-operand MFMA conv path (``ops.set_conv_dtype("f16")``), fp32 tensors / accumulators /
InstanceNorm statistics / losses.  The reference has no AMP: the oracle is the fp32 path, the tolerance is
what fp16 operands allow -- each operand carries 2^-11 relative rounding, a 3x3 conv over C channels averages 9*C such
products -- stated per check below (DESIGN.md "fp16 path").

Gradients are exercised at their REAL magnitude (1e-7 and below at 512^2): without the per-tensor power-of-two scale of
``smsut_absmax_scale`` they are fp16 subnormals / zeros."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, l2_rel
from oracle import recipe, smsut_oracle as O

pytestmark = pytest.mark.gpu
FWD_TOL = 4e-3        # max-norm relative, one block (two stacked 3x3 convs + IN)
# l2-relative, gradients through a block.  Dominated not by the fp16 products but by LeakyReLU mask flips: the forward
# activations move by ~1e-3 relative, so the ~0.1 % of pre-activations that close to zero change sign and re-route their
# gradient (factor 100 between the two slopes) -- the same mechanism that separates the reference's own fp32 and fp64
# gradients, at fp16 instead of fp32 rounding.
GRAD_TOL = 4e-2
# The per-channel InstanceNorm gradients (g, b) are totals of the activation gradient over N*H*W positions with random signs:
# the resultant is ~sqrt(positions) of the summed magnitude while every flipped position contributes its full re-routed
# value, so the same flips weigh more.  Measured over this file's cases (scratch/f16_sc_err.py): 0.009-0.037 with the
# shortcut conv on fp32 operands, 0.009-0.041 with it on fp16 operands inside conv1's pass (one more rounded operand
# feeding the residual tail's sign) -- the worst is IN1's beta at (5, 128, 16, 32) in both.
VEC_TOL = 6e-2


@pytest.fixture()
def ops():
    import smsut_amd  # noqa: F401
    from smsut_amd import ops as o
    prev = o.conv_dtype()
    yield o
    o.set_conv_dtype(prev)


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape) * scale).float()


def hwio(ops, w):
    out = ops.new_weight(*w.shape, device="cuda")
    out.copy_(w)
    return out


def test_absmax_scale(ops):
    from smsut_amd import _hip as H
    for mx, n in ((3.1e-7, 100_003), (1.0, 64), (7000.0, 1 << 20), (0.0, 4096)):
        x = rnd(n, seed=3) * 0.2
        x = x / x.abs().max() * mx if mx else torch.zeros(n)
        out = ops._grad_scale(x.cuda()).cpu().numpy()
        if mx == 0.0:
            assert tuple(out) == (1.0, 1.0)
            continue
        assert out[0] * out[1] == 1.0 and np.log2(out[0]) == np.round(np.log2(out[0]))       # an exact power of two
        assert 2.0 ** 13 <= mx * out[0] <= 2.0 ** 14, (mx, out)


def _block(ops, x, ws_, gout, mode, parts=None):
    """Fused BasicBlock forward + backward in the given operand dtype; returns (out, grads...)."""
    ops.set_conv_dtype(mode)
    w1, g1, b1, w2, g2, b2, ws, gs, bs = [t.clone().requires_grad_(True) if t is not None else None for t in ws_]
    if parts is None:
        xd = x.clone().requires_grad_(True)
        out = ops.basic_block(xd, w1, g1, b1, w2, g2, b2, ws, gs, bs, 0.01)
        leaves = [xd]
    else:
        a, b = [p.clone().requires_grad_(True) for p in parts]
        out = ops.basic_block_cat(ops.CatParts(a, b), w1, g1, b1, w2, g2, b2, ws, gs, bs, 0.01)
        leaves = [a, b]
    out.backward(gout)
    leaves += [t for t in (w1, g1, b1, w2, g2, b2, ws, gs, bs) if t is not None]
    return out.detach(), [t.grad.detach() for t in leaves]


@pytest.mark.parametrize("n,h,ci,co", [(16, 32, 32, 64), (2, 16, 64, 32), (3, 24, 16, 16),          # per-tile kernels, ragged tiles
                                       (8, 128, 16, 16), (5, 128, 16, 32), (4, 128, 64, 32), (9, 64, 32, 64),   # persistent forms
                                       (2, 512, 16, 16), (2, 256, 8, 16)])                        # the config-5 top level; first block after the stem
def test_f16_block_matches_fp32_block(ops, n, h, ci, co):
    """Fused BasicBlock with fp16 operands vs the SAME kernels with fp32 operands (themselves pinned against torch autograd
    in test_ops_gpu.py): output at FWD_TOL, every gradient at GRAD_TOL, with upstream gradients of magnitude 1e-7."""
    x = rnd(n, ci, h, h, seed=1).cuda().contiguous(memory_format=torch.channels_last)
    has_sc = ci != co
    ws_ = [hwio(ops, rnd(co, ci, 3, 3, seed=2) / np.sqrt(9 * ci)), (1 + 0.1 * rnd(co, seed=3)).cuda(), (0.1 * rnd(co, seed=4)).cuda(),
           hwio(ops, rnd(co, co, 3, 3, seed=5) / np.sqrt(9 * co)), (1 + 0.1 * rnd(co, seed=6)).cuda(), (0.1 * rnd(co, seed=7)).cuda(),
           hwio(ops, rnd(co, ci, 1, 1, seed=8) / np.sqrt(ci)) if has_sc else None,
           (1 + 0.1 * rnd(co, seed=9)).cuda() if has_sc else None, (0.1 * rnd(co, seed=10)).cuda() if has_sc else None]
    gout = (rnd(n, co, h, h, seed=11) * 3e-7).cuda().contiguous(memory_format=torch.channels_last)
    o32, g32 = _block(ops, x, ws_, gout, "f32")
    o16, g16 = _block(ops, x, ws_, gout, "f16")
    assert not torch.equal(o32, o16)                                   # the fp16 kernels really ran
    assert rel_err(o16.cpu().numpy(), o32.cpu().numpy()) < FWD_TOL
    for a, b in zip(g16, g32):
        assert float(b.abs().max()) > 0
        err = l2_rel(a.cpu().numpy(), b.cpu().numpy())
        assert err < (VEC_TOL if b.dim() == 1 else GRAD_TOL), (tuple(b.shape), err)
        if b.dim() == 4 and b.shape[0] == n:
            # the activation gradient: its BULK is fp16-accurate, the l2 figure is carried by the few flipped positions
            d = (a - b).abs() / b.abs().max()
            assert float(d.median()) < 2e-3, (tuple(b.shape), float(d.median()))


@pytest.mark.parametrize("n,h,ca,co", [(8, 128, 16, 16), (4, 64, 32, 32), (2, 32, 64, 64), (2, 512, 16, 16)])
def test_f16_block_after_virtual_cat(ops, n, h, ca, co):
    """The decoder form: conv1 / shortcut read cat([up, skip]) in place, the block-input gradient is written into the two
    parts (virtual-cat forward, split-output data-gradient, virtual-cat weight gradient) -- fp16 vs fp32 operands."""
    ci = 2 * ca
    a = rnd(n, ca, h, h, seed=1).cuda().contiguous(memory_format=torch.channels_last)
    b = rnd(n, ca, h, h, seed=2).cuda().contiguous(memory_format=torch.channels_last)
    ws_ = [hwio(ops, rnd(co, ci, 3, 3, seed=2) / np.sqrt(9 * ci)), (1 + 0.1 * rnd(co, seed=3)).cuda(), (0.1 * rnd(co, seed=4)).cuda(),
           hwio(ops, rnd(co, co, 3, 3, seed=5) / np.sqrt(9 * co)), (1 + 0.1 * rnd(co, seed=6)).cuda(), (0.1 * rnd(co, seed=7)).cuda(),
           hwio(ops, rnd(co, ci, 1, 1, seed=8) / np.sqrt(ci)), (1 + 0.1 * rnd(co, seed=9)).cuda(), (0.1 * rnd(co, seed=10)).cuda()]
    if not ops.basic_block_cat_fusable(ops.CatParts(a, b), ws_[0], ws_[6]):
        pytest.skip("virtual-cat kernels do not cover this shape")
    gout = (rnd(n, co, h, h, seed=11) * 1e-6).cuda().contiguous(memory_format=torch.channels_last)
    o32, g32 = _block(ops, None, ws_, gout, "f32", parts=(a, b))
    o16, g16 = _block(ops, None, ws_, gout, "f16", parts=(a, b))
    assert not torch.equal(o32, o16)
    assert rel_err(o16.cpu().numpy(), o32.cpu().numpy()) < FWD_TOL
    for x16, x32 in zip(g16, g32):
        err = l2_rel(x16.cpu().numpy(), x32.cpu().numpy())
        assert err < GRAD_TOL, (tuple(x32.shape), err)


def test_f16_generic_conv_family_first_order_only(ops):
    """Generic conv / dgrad / wgrad Functions (the discriminator's BottleBlock convs): fp16 operands inside
    ``first_order_pass()``, fp32 operands outside it (the WGAN-GP x_hat pass is differentiated twice)."""
    x = rnd(4, 32, 32, 32, seed=1).cuda().contiguous(memory_format=torch.channels_last)
    w = hwio(ops, rnd(64, 32, 3, 3, seed=2) / np.sqrt(9 * 32))
    gy = (rnd(4, 64, 32, 32, seed=3) * 1e-7).cuda().contiguous(memory_format=torch.channels_last)

    def run(first_order):
        xd, wd = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        if first_order:
            with ops.first_order_pass():
                y = ops.conv2d(xd, wd, None, 1, 1)
        else:
            y = ops.conv2d(xd, wd, None, 1, 1)
        y.backward(gy)
        return y.detach(), xd.grad, wd.grad
    ops.set_conv_dtype("f32")
    ref = run(True)
    ops.set_conv_dtype("f16")
    got = run(True)
    keep = run(False)
    for a, b in zip(keep, ref):
        assert torch.equal(a, b)                                       # outside first_order_pass(): untouched fp32 kernels
    assert not torch.equal(got[0], ref[0])
    assert rel_err(got[0].cpu().numpy(), ref[0].cpu().numpy()) < 2e-3
    assert l2_rel(got[1].cpu().numpy(), ref[1].cpu().numpy()) < 2e-3 and l2_rel(got[2].cpu().numpy(), ref[2].cpu().numpy()) < 2e-3


def test_f16_unet_512_vs_fp32_oracle(ops):
    """U-Net(1,5,16) at 512x512 (config 5's slice size), fp16 conv operands + fp16 storage of the block-internal tensors, vs the fp32
    CPU oracle: logits within 1e-2 of the logit range (~40 stacked convs, each 2^-11 per operand; measured 4.3e-3 with half storage,
    2.2e-3 without -- scratch/f16_unet512_vals.py), Dice+CE loss within 1e-4 (measured 3e-6).

    Parameter gradients: the network is not smooth (4 MaxPools, 18 LeakyReLUs), so a forward perturbation of relative size d
    flips a fraction ~d of the argmax / sign decisions and moves the gradient by ~sqrt(d) in l2 terms.  SURVEY.md section 9
    measured that law on the reference itself: fp32 vs fp64 (d ~ 1e-7) -> median 1.4e-3, worst 6.5e-3.  fp16 operands are
    d ~ 5e-4, i.e. sqrt(5000) ~ 70x that: ~0.1.  Bars: median < 0.12, worst < 0.2 (measured 0.110 / 0.169 with half storage, 0.092 /
    0.162 without), and the direction is kept -- cosine similarity of the full gradient vector > 0.995 (measured 0.9991)."""
    from smsut_amd.network.unet import UNet
    from smsut_amd.misc.loss import DiceAndCrossEntropyLoss
    torch.set_num_threads(16)
    H = 512
    x = recipe.synth_images((2, 1, H, H), 501)
    y = recipe.synth_labels(2, H, H, 5, 502)
    sd = recipe.fill(recipe.unet_shapes(1, 5, 16), 500)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.unet_forward(leaf, x)
    ref_loss = O.dice_ce(ref, y)
    ref_loss.backward()
    ops.set_conv_dtype("f16")
    net = UNet(1, 5, 16, norm_type="instance", act_type="lrelu")
    net.load_state_dict(sd); net.cuda().train()
    out = net(x.cuda())
    loss = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)(out, y.cuda())
    loss.backward()
    e = rel_err(out.detach().cpu().numpy(), ref.detach().numpy())
    assert 1e-5 < e < 1e-2, e                                          # (> 1e-5: the fp16 kernels really ran)
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    errs = {k: l2_rel(p.grad.cpu().numpy(), leaf[k].grad.numpy()) for k, p in net.named_parameters()}
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert float(np.median(list(errs.values()))) < 0.12 and worst[1] < 0.2, (float(np.median(list(errs.values()))), worst)
    ga = torch.cat([p.grad.detach().cpu().double().reshape(-1) for _, p in net.named_parameters()])
    gb = torch.cat([leaf[k].grad.double().reshape(-1) for k, _ in net.named_parameters()])
    cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
    assert cos > 0.995, cos


def test_f16_ugan_consis_iteration_512_vs_fp32_oracle(ops):
    """One uganConsis iteration at config 5's size (512x512, 1 labeled + 1 unlabeled slice, the 7-stage discriminator of
    ugan.py:205-215) with fp16 conv operands and half storage vs the fp32 CPU oracle, optimizers at lr 0: segmentor-side scalars at
    1e-3 (measured <= 8.3e-5, scratch/f16_iter512_vals.py), everything else (through the tanh translator and D) at 1.5e-2 (measured
    <= 5.0e-3: G_cls), the gradient penalty at 2e-2 (7.2e-3).  (r03's bars were 1e-2 / 5e-2 / 5e-2.)"""
    from smsut_amd import config as cfg
    from smsut_amd.trainer.uganConsisTrainer import UGANConsisTrainer, SCALARS
    old = (cfg.input_size, cfg.batch_size)
    cfg.input_size, cfg.batch_size = 512, 1
    try:
        ops.set_conv_dtype("f16")
        tr = UGANConsisTrainer("train", types.SimpleNamespace(fold=0, expr_name=None, write_env=False))
        g_w = recipe.fill(recipe.ugan_shapes(1, 5, 4, 16), 91)
        d_w = recipe.fill(recipe.disc_shapes(512, 4, 16, 256), 92)
        tr.net.load_state_dict(g_w); tr.D.load_state_dict(d_w)
        tr.net.train(); tr.D.train()
        tr.epoch, tr.iter = 100, 15000
        for grp in list(tr.d_optimizer.param_groups) + list(tr.optimizer.param_groups):
            grp["lr"] = 0.0
        x, y, modal, mj, alpha, ids = recipe.trace_inputs(0, b=2, size=512, base=1500)
        got = np.array(tr.train_iteration(x.cuda(), y.cuda(), modal, mj=mj, alpha=alpha.cuda(), sample_ids=[ids.cuda()]).tolist())
        torch.set_num_threads(16)
        gsd = {k: v.clone().requires_grad_(True) for k, v in g_w.items()}
        dsd = {k: v.clone().requires_grad_(True) for k, v in d_w.items()}
        logs, _ = O.ugan_consis_iteration(gsd, dsd, torch.optim.SGD(list(gsd.values()), lr=0.0),
                                          torch.optim.Adam(list(dsd.values()), 0.0), x, y, modal, mj, alpha, [ids], it=15000,
                                          epoch=100, nce_batch=1, base_lr=0.0)
        ref = np.array([logs[k] for k in SCALARS])
        rep = dict(zip(SCALARS, zip(got, ref)))
        assert np.isfinite(got).all(), rep
        seg = [SCALARS.index(k) for k in ("G_seg", "G_semi", "G_rec")]
        assert np.allclose(got[seg], ref[seg], rtol=1e-3, atol=1e-4), rep
        rest = [i for i in range(10) if i not in seg and SCALARS[i] != "D_gp"]
        assert np.allclose(got[rest], ref[rest], rtol=1.5e-2, atol=2e-3), rep
        i_gp = SCALARS.index("D_gp")                                   # the x_hat pass keeps fp32 operands; x_fake itself is f16-made
        assert abs(got[i_gp] - ref[i_gp]) <= 2e-2 * abs(ref[i_gp]), rep
    finally:
        cfg.input_size, cfg.batch_size = old


@pytest.mark.parametrize("n,h,ci,co,cat", [(4, 128, 16, 32, 0), (12, 128, 32, 16, 1), (16, 64, 64, 32, 1), (2, 256, 16, 16, 0), (8, 64, 64, 64, 0)])
def test_fused_shortcut_forward_with_fp16_operands(ops, n, h, ci, co, cat):
    """conv1 + the block's 1x1 shortcut in one pass with fp16 operands (config 5; network/blocks.py:66-80): the 3x3 half is
    BIT-IDENTICAL to the unfused fp16-operand entry point (same kernel, same order), the shortcut half matches fp64 at what fp16
    operands allow (each carries 2^-11 relative rounding; the sum runs over ci products), InstanceNorm partials of both outputs."""
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    assert H.call("smsut_conv2d_fwd_sc_f16_supported", n, h, h, ci, co, cat) == 1
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(n, h, h, ci, generator=g).cuda()
    w3 = (torch.randn(9 * ci * co, generator=g) / np.sqrt(9 * ci)).cuda()
    w1 = (torch.randn(ci * co, generator=g) / np.sqrt(ci)).cuda()
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, 1)
    y0 = torch.full((n, h, h, co), float("nan"), device="cuda"); p0 = torch.zeros(n * tiles * co * 2, device="cuda")
    if cat:
        xa, xb = x[..., :ci // 2].contiguous(), x[..., ci // 2:].contiguous()
        H.call("smsut_conv2d_fwd_mfma_stats_cat_f16", xa, xb, w3, y0, p0, n, h, h, ci, co, st)
    else:
        xa, xb = x, None
        H.call("smsut_conv2d_fwd_mfma_stats_f16", x, w3, y0, p0, n, h, h, ci, co, 3, st)
    y1, s1 = torch.full_like(y0, float("nan")), torch.full_like(y0, float("nan"))
    p1, q1 = torch.zeros_like(p0), torch.zeros_like(p0)
    H.call("smsut_conv2d_fwd_mfma_stats_sc_f16", xa, xb, w3, w1, y1, s1, p1, q1, n, h, h, ci, co, st)
    assert torch.equal(y1, y0) and torch.equal(p1, p0)
    ref = (x.double().reshape(-1, ci) @ w1.double().view(ci, co)).reshape(n, h, h, co)
    assert float((s1.double() - ref).abs().max() / ref.abs().max()) < 2e-3
    q = q1.view(n, tiles, co, 2).double().sum(1)
    assert torch.allclose(q[..., 0], s1.double().sum((1, 2)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(q[..., 1], (s1.double() ** 2).sum((1, 2)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("n,h,co,ci,split", [(4, 128, 16, 32, 0), (8, 128, 16, 32, 16), (4, 128, 32, 64, 32), (2, 256, 16, 16, 0),
                                              (16, 64, 32, 32, 16)])
def test_fused_shortcut_data_gradient_with_fp16_operands(ops, n, h, co, ci, split):
    """gx = dgrad3x3(gy1, w1) + dgrad1x1(gs, ws) in ONE pass with fp16 operands and ONE power-of-two scale over both gradients
    (``smsut_absmax_scale2``), split output for blocks after a concat (network/blocks.py:66-80).  Against fp64 at what fp16
    operands allow: each operand carries 2^-11 relative rounding, the sums run over 9 co + co products (bound: max-norm 2e-3)."""
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    assert H.call("smsut_conv2d_dgrad_sc_f16_supported", n, h, h, co, ci, split) == 1
    g = torch.Generator(device="cpu").manual_seed(11)
    gy = (torch.randn(n, h, h, co, generator=g) * 2e-7).cuda()
    gs = (torch.randn(n, h, h, co, generator=g) * 5e-8).cuda()
    w3 = (torch.randn(3, 3, ci, co, generator=g) / np.sqrt(9 * ci)).cuda()          # forward layouts [kh][kw][Cin][Cout], [Cin][Cout]
    w1 = (torch.randn(ci, co, generator=g) / np.sqrt(ci)).cuda()
    sc = torch.empty(2, device="cuda")
    H.call("smsut_absmax_scale2", gy, gy.numel(), gs, gs.numel(), sc, torch.empty(1024, device="cuda"), st)
    m = max(float(gy.abs().max()), float(gs.abs().max()))
    assert 2.0 ** 13 <= m * float(sc[0]) <= 2.0 ** 14 and float(sc[0] * sc[1]) == 1.0
    if split:
        ga = torch.full((n, h, h, split), float("nan"), device="cuda")
        gb = torch.full((n, h, h, ci - split), float("nan"), device="cuda")
        H.call("smsut_conv2d_dgrad_mfma_sc_f16", gy, gs, w3, w1, ga, gb, sc, split, n, h, h, co, ci, st)
        got = torch.cat([ga, gb], 3)
    else:
        got = torch.full((n, h, h, ci), float("nan"), device="cuda")
        H.call("smsut_conv2d_dgrad_mfma_sc_f16", gy, gs, w3, w1, got, None, sc, 0, n, h, h, co, ci, st)
    wt = w3.double().permute(3, 2, 0, 1)                                            # [Cout, Cin, 3, 3]
    ref = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1)
    ref = ref + (gs.double().reshape(-1, co) @ w1.double().t()).reshape(n, h, h, ci)
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2e-3


@pytest.mark.parametrize("n,h,ci,co,cat", [(4, 64, 16, 32, 0), (4, 64, 32, 16, 1), (2, 128, 32, 64, 0), (3, 64, 64, 32, 1),
                                           (2, 64, 16, 16, 0), (2, 32, 128, 64, 1)])
def test_fused_shortcut_weight_gradient_with_fp16_operands(ops, n, h, ci, co, cat):
    """conv1's weight gradient + the 1x1 shortcut's in one pass over x with fp16 operands (10 tap rows).  Rows 0..8 are
    BIT-IDENTICAL to the unfused fp16-operand weight gradient run with the same scale (same kernel, same accumulators, same
    slab order); row 9 matches fp64 at fp16-operand accuracy (sum over N*H*W products with random signs: l2-relative 2e-3)."""
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    assert H.call("smsut_conv2d_wgrad_sc_f16_supported", n, h, h, ci, co) == 1
    g = torch.Generator(device="cpu").manual_seed(13)
    x = torch.randn(n, h, h, ci, generator=g).cuda()
    gy = (torch.randn(n, h, h, co, generator=g) * 2e-7).cuda()
    gs = (torch.randn(n, h, h, co, generator=g) * 6e-7).cuda()
    sc = torch.empty(2, device="cuda")
    H.call("smsut_absmax_scale2", gy, gy.numel(), gs, gs.numel(), sc, torch.empty(1024, device="cuda"), st)
    if cat:
        xa, xb, ca = x[..., :ci // 2].contiguous(), x[..., ci // 2:].contiguous(), ci // 2
    else:
        xa, xb, ca = x, None, 0
    g9 = torch.full((9 * ci * co,), float("nan"), device="cuda")
    H.call("smsut_conv2d_wgrad_f16", xa, xb, ca, gy, g9, torch.empty(H.call("smsut_conv2d_wgrad_f16_ws", n, h, h, ci, co), device="cuda"),
           sc, n, h, h, ci, co, st)
    g10 = torch.full((10 * ci * co,), float("nan"), device="cuda")
    H.call("smsut_conv2d_wgrad_sc_f16", xa, xb, ca, gy, gs, g10,
           torch.empty(H.call("smsut_conv2d_wgrad_sc_f16_ws", n, h, h, ci, co), device="cuda"), sc, n, h, h, ci, co, st)
    assert torch.equal(g10[:9 * ci * co], g9)
    ref = x.double().reshape(-1, ci).t() @ gs.double().reshape(-1, co)             # [Cin][Cout]
    got = g10[9 * ci * co:].view(ci, co).double()
    assert float((got - ref).norm() / ref.norm()) < 2e-3


@pytest.mark.parametrize("n,h,ci,co", [(5, 128, 16, 32), (8, 128, 16, 16), (4, 128, 64, 32)])
def test_gradient_scales_from_producer_maxima_are_the_absmax_pass(ops, monkeypatch, n, h, ci, co):
    """fp16 operands: ``smsut_restail_bwd_amax`` / ``smsut_in_apply_bwd_amax`` hand max|gy2|, max|gs|, max|gy1| to
    ``smsut_absmax_finish`` instead of one ``smsut_absmax_scale`` pass per gradient tensor -- the same maxima, hence the same
    power-of-two scales: every block gradient is BIT-IDENTICAL to the run with the separate passes."""
    x = rnd(n, ci, h, h, seed=1).cuda().contiguous(memory_format=torch.channels_last)
    has_sc = ci != co
    ws_ = [hwio(ops, rnd(co, ci, 3, 3, seed=2) / np.sqrt(9 * ci)), (1 + 0.1 * rnd(co, seed=3)).cuda(), (0.1 * rnd(co, seed=4)).cuda(),
           hwio(ops, rnd(co, co, 3, 3, seed=5) / np.sqrt(9 * co)), (1 + 0.1 * rnd(co, seed=6)).cuda(), (0.1 * rnd(co, seed=7)).cuda(),
           hwio(ops, rnd(co, ci, 1, 1, seed=8) / np.sqrt(ci)) if has_sc else None,
           (1 + 0.1 * rnd(co, seed=9)).cuda() if has_sc else None, (0.1 * rnd(co, seed=10)).cuda() if has_sc else None]
    gout = (rnd(n, co, h, h, seed=11) * 3e-7).cuda().contiguous(memory_format=torch.channels_last)
    from smsut_amd import profiling
    res = {}
    rec = profiling.record_step(lambda: res.update(r=_block(ops, x, ws_, gout, "f16")))
    o1, g1 = res["r"]
    names = [r[0] for r in rec]
    tail = "smsut_restail_bwd_hs" if "smsut_restail_bwd_hs" in names else "smsut_restail_bwd_amax"      # (half storage where it applies)
    assert tail in names and ("smsut_in_apply_bwd_hs" in names or "smsut_in_apply_bwd_amax" in names)
    assert "smsut_absmax_finish" in names and "smsut_absmax_scale" not in names and "smsut_absmax_scale2" not in names
    monkeypatch.setattr(ops, "AMAX_HANDOVER", False)
    rec = profiling.record_step(lambda: res.update(r=_block(ops, x, ws_, gout, "f16")))
    o0, g0 = res["r"]
    assert "smsut_absmax_scale" in [r[0] for r in rec]
    assert torch.equal(o0, o1)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)


# ---- fp16 STORAGE of the block-internal raw conv outputs (y1, y2, s): "_hs" entry points ------------------------------------
def _hs_shapes():
    return [(4, 128, 16, 32, 0), (12, 128, 32, 16, 1), (2, 256, 16, 16, 0), (8, 64, 64, 64, 0), (2, 256, 8, 16, 0)]


@pytest.mark.parametrize("n,h,ci,co,cat", _hs_shapes())
def test_half_storage_conv_epilogues_round_the_fp32_result(ops, n, h, ci, co, cat):
    """``smsut_conv2d_fwd_mfma_stats_f16_hs`` / ``_sc_f16_hs`` store EXACTLY the fp16 rounding of what the fp32-storage entry
    points store (same kernel, same accumulators; round-to-nearest-even at the store); InstanceNorm partials are bit-identical
    (taken from the fp32 accumulators)."""
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    assert H.call("smsut_conv2d_f16_hs_supported", n, h, h, ci, co, cat) == 1
    g = torch.Generator(device="cpu").manual_seed(17)
    x = torch.randn(n, h, h, ci, generator=g).cuda()
    w3 = (torch.randn(9 * ci * co, generator=g) / np.sqrt(9 * ci)).cuda()
    w1 = (torch.randn(ci * co, generator=g) / np.sqrt(ci)).cuda()
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, ci, co, 3, int(ci != 8))
    xa, xb = (x[..., :ci // 2].contiguous(), x[..., ci // 2:].contiguous()) if cat else (x, None)
    y0, s0 = torch.empty(n, h, h, co, device="cuda"), torch.empty(n, h, h, co, device="cuda")
    p0, q0 = torch.zeros(n * tiles * co * 2, device="cuda"), torch.zeros(n * tiles * co * 2, device="cuda")
    # (8 input channels -- the first block after the stem: fp32 operands, there is no fp16 twin of the 8-channel form)
    H.call("smsut_conv2d_fwd_mfma_stats_sc_f16" if ci != 8 else "smsut_conv2d_fwd_mfma_stats_sc", xa, xb, w3, w1, y0, s0, p0, q0,
           n, h, h, ci, co, st)
    yh = torch.full((n, h, h, co), float("nan"), device="cuda", dtype=torch.float16)
    sh = torch.full_like(yh, float("nan"))
    p1, q1 = torch.zeros_like(p0), torch.zeros_like(p0)
    H.call("smsut_conv2d_fwd_mfma_stats_sc_f16_hs", xa, xb, w3, w1, yh, sh, p1, q1, n, h, h, ci, co, st)
    assert torch.equal(yh, y0.half()) and torch.equal(sh, s0.half()) and torch.equal(p1, p0) and torch.equal(q1, q0)
    if ci != 8:
        yh2 = torch.full_like(yh, float("nan"))
        p2 = torch.zeros_like(p0)
        H.call("smsut_conv2d_fwd_mfma_stats_f16_hs", xa, xb, w3, yh2, p2, n, h, h, ci, co, st)
        assert torch.equal(yh2, yh) and torch.equal(p2, p0)


@pytest.mark.parametrize("n,h,c", [(4, 128, 32), (2, 256, 16), (8, 64, 64)])
def test_half_storage_readers_equal_the_fp32_readers_on_the_same_values(ops, n, h, c):
    """Every kernel that READS a block-internal tensor in half storage (IN apply forward / backward, both residual-tail passes,
    the BST data-gradient's mask read) computes in fp32 from the fp16 values: BIT-IDENTICAL to the fp32-storage entry point fed
    the same values widened to fp32."""
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    hw = h * h
    g = torch.Generator(device="cpu").manual_seed(19)
    R = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).cuda()
    y1h, y2h, sh_ = R(n, h, h, c).half(), R(n, h, h, c).half(), R(n, h, h, c).half()
    y1f, y2f, sf = y1h.float(), y2h.float(), sh_.float()
    gam = [1 + 0.1 * R(c) for _ in range(3)]
    bet = [0.1 * R(c) for _ in range(3)]
    st_ = lambda: (R(n, c, sc=0.1), (1 + 0.1 * R(n, c)).abs())
    (m1, r1), (m2, r2), (ms, rs) = st_(), st_(), st_()
    E = lambda *sh: torch.full(sh, float("nan"), device="cuda")
    # IN apply forward (statistics from partials of one chunk)
    part = torch.stack([y1f.sum((1, 2)), (y1f * y1f).sum((1, 2))], -1).reshape(n, 1, c, 2).contiguous()
    outs = []
    for name, src in (("smsut_instnorm_fwd_partials", y1f), ("smsut_instnorm_fwd_partials_hs", y1h)):
        a, mm, rr = E(n, h, h, c), E(n, c), E(n, c)
        H.call(name, src, gam[0], bet[0], a, mm, rr, part, 1, n, hw, c, 1e-5, 0.01, 1, st)
        outs.append((a, mm, rr))
    assert all(torch.equal(p, q) for p, q in zip(*outs))
    a1f = outs[0][0]
    a1h = torch.full((n, h, h, c), float("nan"), device="cuda", dtype=torch.float16)
    H.call("smsut_instnorm_fwd_partials_hs2", y1h, gam[0], bet[0], a1h, E(n, c), E(n, c), part, 1, n, hw, c, 1e-5, 0.01, 1, st)
    assert torch.equal(a1h, a1f.half())                     # the activated tensor, rounded at the store
    if H.call("smsut_conv2d_f16_hs_supported", n, h, h, c, c, 0):
        # conv2 forward and its weight gradient read that fp16 a1: the operand bits the fp32-input forms round to -> same results
        w = (R(9 * c * c) / np.sqrt(9 * c))
        tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, c, c, 3, 1)
        ya, yb = (torch.full((n, h, h, c), float("nan"), device="cuda", dtype=torch.float16) for _ in range(2))
        pa, pb_ = (torch.zeros(n * tiles * c * 2, device="cuda") for _ in range(2))
        H.call("smsut_conv2d_fwd_mfma_stats_f16_hs", a1f, None, w, ya, pa, n, h, h, c, c, st)
        H.call("smsut_conv2d_fwd_mfma_stats_f16_hsx", a1h, w, yb, pb_, n, h, h, c, c, st)
        assert torch.equal(ya, yb) and torch.equal(pa, pb_)
        gy = R(n, h, h, c, sc=1e-7)
        sc = torch.empty(2, device="cuda")
        H.call("smsut_absmax_scale", gy, gy.numel(), sc, torch.empty(1024, device="cuda"), st)
        wsz = H.call("smsut_conv2d_wgrad_f16_ws", n, h, h, c, c)
        ga, gb_ = E(9 * c * c), E(9 * c * c)
        H.call("smsut_conv2d_wgrad_f16", a1f, None, 0, gy, ga, torch.empty(wsz, device="cuda"), sc, n, h, h, c, c, st)
        H.call("smsut_conv2d_wgrad_f16_xh", a1h, gy, gb_, torch.empty(wsz, device="cuda"), sc, n, h, h, c, c, st)
        assert torch.equal(ga, gb_)
        # ... or read the RAW fp16 y1 and normalise + activate it while staging (a1 never built): the same operand bits again
        mm, rr = outs[0][1], outs[0][2]
        yc, pc, gc = torch.full_like(ya, float("nan")), torch.zeros_like(pa), E(9 * c * c)
        H.call("smsut_conv2d_fwd_mfma_stats_inaff_f16_hsx", y1h, w, yc, pc, mm, rr, gam[0], bet[0], 0.01, n, h, h, c, c, st)
        assert torch.equal(yc, ya) and torch.equal(pc, pa)
        H.call("smsut_conv2d_wgrad_f16_xh_inaff", y1h, gy, gc, torch.empty(wsz, device="cuda"), sc, mm, rr, gam[0], bet[0], 0.01,
               n, h, h, c, c, st)
        assert torch.equal(gc, ga)
    # residual tail forward
    outs = []
    for name, a, b in (("smsut_restail_fwd", y2f, sf), ("smsut_restail_fwd_hs", y2h, sh_)):
        o = E(n, h, h, c)
        H.call(name, a, m2, r2, gam[1], bet[1], b, ms, rs, gam[2], bet[2], o, n, hw, c, 0.01, st)
        outs.append(o)
    assert torch.equal(*outs)
    out = outs[0]
    # residual tail backward (+ the maxima hand-over)
    gout = R(n, h, h, c, sc=1e-7)
    chunks = H.call("smsut_in_chunks", n, hw, c)
    nb = H.call("smsut_amax_blocks", n, hw, c)
    res = []
    for hs in (False, True):
        gy2, gs_t = E(n, h, h, c), E(n, h, h, c)
        vecs = [E(n, c) for _ in range(3)] + [E(c) for _ in range(4)]
        amax = E(2 * nb)
        args = [gout, out, y2h if hs else y2f, m2, r2, gam[1], bet[1], sh_ if hs else sf, ms, rs, gam[2], bet[2], gy2, gs_t, *vecs,
                torch.empty(n * chunks * c * 3, device="cuda"), amax, n, hw, c, 0.01, st]
        H.call("smsut_restail_bwd_hs" if hs else "smsut_restail_bwd_amax", *args)
        res.append([gy2, gs_t, *vecs, amax])
    assert all(torch.equal(p, q) for p, q in zip(*res))
    assert float(res[0][-1][:nb].max()) == float(res[0][0].abs().max()) and float(res[0][-1][nb:].max()) == float(res[0][1].abs().max())
    # IN apply backward
    gz, am, bm = R(n, h, h, c, sc=1e-7), R(n, c, sc=1e-9), R(n, c, sc=1e-9)
    res = []
    for hs in (False, True):
        gx, gg, gb, amax = E(n, h, h, c), E(c), E(c), E(nb)
        H.call("smsut_in_apply_bwd_hs" if hs else "smsut_in_apply_bwd_amax", gz, y1h if hs else y1f, m1, r1, gam[0], am, bm, gx, gg, gb, amax,
               n, hw, c, st)
        res.append([gx, gg, gb, amax])
    assert all(torch.equal(p, q) for p, q in zip(*res))
    # conv2's data-gradient with the IN1 / LeakyReLU mask and backward partials
    if H.call("smsut_conv2d_f16_hs_supported", n, h, h, c, c, 0):
        w = (R(9 * c * c) / np.sqrt(9 * c))
        sc = torch.empty(2, device="cuda")
        H.call("smsut_absmax_scale", gout, gout.numel(), sc, torch.empty(1024, device="cuda"), st)
        tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, c, c, 3, 1)
        res = []
        for hs in (False, True):
            gz2, pb = E(n, h, h, c), torch.zeros(n * tiles * c * 2, device="cuda")
            H.call("smsut_conv2d_dgrad_mfma_bwdstats_f16_hs" if hs else "smsut_conv2d_dgrad_mfma_bwdstats_f16", gout, w, gz2, pb,
                   y1h if hs else y1f, m1, r1, gam[0], bet[0], sc, 0.01, n, h, h, c, c, st)
            res.append([gz2, pb])
        assert all(torch.equal(p, q) for p, q in zip(*res))


def test_gradient_scale_entry_points_edge_cases(ops):
    """``smsut_absmax_scale`` / ``_scale2`` / ``smsut_absmax_finish``: the scale is the power of two that puts the maximum into
    [2^13, 2^14]; an all-zero operand and a non-finite maximum give the neutral pair {1, 1}; ``_scale2`` is the scale of the larger of
    two tensors; ``_finish`` over per-workgroup maxima is the scale of their maximum."""
    from smsut_amd import _hip as H
    st = H.stream_ptr()
    ws = torch.empty(1024, device="cuda")
    out = torch.empty(2, device="cuda")

    def scale(t):
        H.call("smsut_absmax_scale", t, t.numel(), out, ws, st)
        return out.tolist()
    x = torch.randn(100003, device="cuda") * 3e-7                         # (odd length: the scalar tail path)
    s, si = scale(x)
    m = float(x.abs().max())
    assert 2.0 ** 13 <= m * s <= 2.0 ** 14 and s * si == 1.0 and np.log2(s) == round(np.log2(s))
    assert scale(torch.zeros(4096, device="cuda")) == [1.0, 1.0]
    bad = x.clone(); bad[17] = float("inf")
    assert scale(bad) == [1.0, 1.0]
    y = torch.randn(4096, device="cuda") * 5e-5
    H.call("smsut_absmax_scale2", x[:100000], 100000, y, y.numel(), out, ws, st)
    assert out.tolist() == scale(y)                                       # y holds the larger maximum
    slots = torch.tensor([0.0, 3e-7, 1.25e-6, 0.0, 9e-7], device="cuda")
    H.call("smsut_absmax_finish", slots, slots.numel(), out, st)
    assert out.tolist() == scale(torch.full((4,), 1.25e-6, device="cuda"))


@pytest.mark.parametrize("n,h,ci,co", [(2, 32, 128, 128), (3, 16, 256, 64), (2, 32, 64, 32), (8, 128, 16, 16), (8, 128, 32, 32), (2, 48, 96, 160)])
def test_fp16_operand_products_are_exact_on_fp16_exact_integers(ops, n, h, ci, co):
    """r05 (v_mfma_f32_16x16x32_f16 in tap pairs / 32-channel passes; a mixed 16- / 32-deep sequence once dropped contributions): with
    small-integer inputs and weights every fp16 conversion, every product and every fp32 partial sum is exact, so the fp16-operand
    forward AND data-gradient entry points must reproduce torch's fp32 convolution bit for bit -- for one-hot single taps (a wrong tap
    pairing or a dropped issue shows as a missing tap) and for full 3x3 kernels, on shapes that take the persistent kernel (>= 1024
    items) and on shapes that take the per-tile kernel with 32-channel passes."""
    import torch.nn.functional as F
    from smsut_amd import _hip as H
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(n * 1000 + ci)
    x = torch.randint(-4, 5, (n, ci, h, h), device=dev, generator=g).float().contiguous(memory_format=torch.channels_last)
    gy = torch.randint(-4, 5, (n, co, h, h), device=dev, generator=g).float().contiguous(memory_format=torch.channels_last)
    st = torch.cuda.current_stream().cuda_stream
    for tap in (0, 4, 7, 8, -1):
        w = torch.zeros(co, ci, 3, 3, device=dev)
        if tap >= 0:
            w[:, :, tap // 3, tap % 3] = torch.randint(-2, 3, (co, ci), device=dev, generator=g).float()
        else:
            w = torch.randint(-2, 3, (co, ci, 3, 3), device=dev, generator=g).float()
        wh = ops.new_weight(co, ci, 3, 3, device=dev)
        wh.copy_(w)
        y = torch.empty(n, co, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        H.call("smsut_conv2d_fwd_mfma_f16", x, wh, y, None, n, h, h, ci, co, 3, 0, st)
        assert torch.equal(y, F.conv2d(x, w, padding=1)), ("forward", tap)
        gx = torch.empty(n, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        H.call("smsut_conv2d_fwd_mfma_f16", gy, wh, gx, None, n, h, h, co, ci, 3, 1, st)
        assert torch.equal(gx, F.conv_transpose2d(gy, w, padding=1)), ("data-gradient", tap)
