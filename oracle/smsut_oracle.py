"""CPU oracle for the SMSUT conv hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional PyTorch (fp32 or fp64, CPU), the arithmetic of
the reference's hot path so that the HIP product path can be checked against it.  It is
imported only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg -- never by anything under ``smsut-medicalimgsegmentation_amd/``.

Pinning: the reference ships no golden vectors, so the oracle is
pinned by fixtures generated from the reference's own modules imported on CPU in the
build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` replays every fixture through this file.

Every network here is a pure function of a ``state_dict``-shaped mapping
``{key: tensor}`` with the reference's key names and OIHW tensor shapes, so the same
weights can be fed to the reference modules, to this oracle and to the HIP modules.

Reference citations are ``path:line`` under the upstream repository root.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

IN_EPS = 1e-5          # nn.InstanceNorm2d default, network/blocks.py:23
LRELU_SLOPE = 1e-2     # network/blocks.py:28 (negative=1e-2); nn.LeakyReLU default, network/ugan.py:203
NCE_T = 0.07           # network/patchnce.py:46
DICE_SMOOTH = 1e-5     # misc/loss.py:40
DICE_EPS = 1e-8        # misc/loss.py:56


# --------------------------------------------------------------------------- blocks
def _inorm(x, sd: SD, p: str):
    """InstanceNorm2d(affine=True), biased variance, no running stats (network/blocks.py:23)."""
    return F.instance_norm(x, weight=sd[p + "weight"], bias=sd[p + "bias"], eps=IN_EPS)


def _act(x, slope=LRELU_SLOPE):
    return F.leaky_relu(x, slope)


def basic_block(sd: SD, p: str, x, slope=LRELU_SLOPE):
    """network/blocks.py:53-80 -- conv3x3-IN-act-conv3x3-IN (+ conv1x1-IN shortcut) add act."""
    y = _act(_inorm(F.conv2d(x, sd[p + "conv1.weight"], padding=1), sd, p + "bn1."), slope)
    y = _inorm(F.conv2d(y, sd[p + "conv2.weight"], padding=1), sd, p + "bn2.")
    if (p + "shortcut1.weight") in sd:
        idn = _inorm(F.conv2d(x, sd[p + "shortcut1.weight"]), sd, p + "shortcut2.")
    else:
        idn = x
    return _act(y + idn, slope)


def bottle_block(sd: SD, p: str, x, stride=2, slope=LRELU_SLOPE):
    """network/blocks.py:83-117 -- the avg-pool sits BETWEEN conv1 and conv2 and on the identity."""
    idn = F.avg_pool2d(x, 2) if stride == 2 else x
    y = _act(_inorm(F.conv2d(x, sd[p + "conv1.weight"], padding=1), sd, p + "bn1."), slope)
    if stride == 2:
        y = F.avg_pool2d(y, 2)
    y = _inorm(F.conv2d(y, sd[p + "conv2.weight"], padding=1), sd, p + "bn2.")
    if (p + "downsample.0.weight") in sd:
        idn = _inorm(F.conv2d(idn, sd[p + "downsample.0.weight"]), sd, p + "downsample.1.")
    return _act(y + idn, slope)


def up_and_concat(sd: SD, p: str, x, skip):
    """network/blocks.py:37-50 -- ConvT 2x2 s2 or bilinear x2 + conv1x1; cat([up, skip])."""
    if (p + "up.weight") in sd:                       # transposed=True
        u = F.conv_transpose2d(x, sd[p + "up.weight"], stride=2)
    else:                                             # nn.Sequential(Upsample, conv1x1) -> key up.1.weight
        u = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        u = F.conv2d(u, sd[p + "up.1.weight"])
    return torch.cat([u, skip], dim=1)


# --------------------------------------------------------------------------- U-Net
def unet_forward(sd: SD, x, slope=LRELU_SLOPE):
    """network/unet.py:29-32 + network/blocks.py:137-152,168-174 (bias-free 1x1 head)."""
    h = _act(_inorm(F.conv2d(x, sd["encoder.pre_conv.weight"], padding=2), sd, "encoder.pre_bn."), slope)
    skips = []
    for lvl in (1, 2, 3, 4):
        h = basic_block(sd, f"encoder.layer{lvl}.", h, slope)
        skips.append(h)
        h = F.max_pool2d(h, 2, 2)
    h = basic_block(sd, "encoder.layer5.", h, slope)
    for lvl in (4, 3, 2, 1):
        h = basic_block(sd, f"decoder.layer{lvl}.", up_and_concat(sd, f"decoder.up{lvl}.", h, skips[lvl - 1]), slope)
    return F.conv2d(h, sd["decoder.fc.weight"])


# --------------------------------------------------------------------------- ugan generator
def _ugan_encoder(sd: SD, p: str, x):
    """network/ugan.py:22-55 -- returns (pooled bottleneck input, skips deepest-first)."""
    h = _act(_inorm(F.conv2d(x, sd[p + "pre.0.weight"], padding=2), sd, p + "pre.1."))
    skips = []
    for lvl in (1, 2, 3, 4):
        h = basic_block(sd, f"{p}enc{lvl}.", h)
        skips.append(h)
        h = F.max_pool2d(h, 2, 2)
    skips.reverse()
    return h, skips


def _ugan_decoder(sd: SD, p: str, e5, skips, use_tanh: bool):
    """network/ugan.py:58-83 -- 1x1 head WITH bias, optional tanh."""
    h = e5
    for i, lvl in enumerate((4, 3, 2, 1)):
        h = basic_block(sd, f"{p}dec{lvl}.", up_and_concat(sd, f"{p}up{lvl}.", h, skips[i]))
    out = F.conv2d(h, sd[p + "fc.weight"], sd[p + "fc.bias"])
    return torch.tanh(out) if use_tanh else out


def l2_normalize(x):
    """network/networks.py:234-243 (power=2): x / (||x||_2 + 1e-7)."""
    norm = x.pow(2).sum(1, keepdim=True).pow(0.5)
    return x / (norm + 1e-7)


def patch_sample(sd: SD, feat, patch_id, p="netF.mlp_0."):
    """network/ugan.py:302-339 for one feature map: shared ids, 2-layer MLP, L2 norm."""
    fr = feat.permute(0, 2, 3, 1).flatten(1, 2)          # [B, HW, C]
    xs = fr[:, patch_id, :].flatten(0, 1)                # [B*P, C]
    xs = F.linear(xs, sd[p + "0.weight"], sd[p + "0.bias"])
    xs = F.linear(F.relu(xs), sd[p + "2.weight"], sd[p + "2.bias"])
    return l2_normalize(xs)


def ugan_forward(sd: SD, x, m=None, sample_ids: Optional[Sequence[torch.Tensor]] = None,
                 val_phase: bool = False, n_modal: int = 4, with_nce: bool = True):
    """network/ugan.py:153-195 (UGANnce) / :108-123 (UGAN when with_nce=False).

    ``sample_ids`` must be given when ``with_nce`` and not ``val_phase``: the oracle never
    draws the ``randperm`` itself (RNG-dependent values are captured and fed in).
    """
    B, _, H, W = x.shape
    if m is None:
        m = torch.zeros(B, n_modal, dtype=x.dtype)
    planes = m.view(B, -1, 1, 1).repeat(1, 1, H, W)      # ugan.py:156-157
    t_in = torch.cat([x, planes], dim=1)                 # ugan.py:159

    t_bot, t_sk = _ugan_encoder(sd, "tsl_encoder.", t_in)
    t_e5 = basic_block(sd, "enc5.", t_bot)               # shared enc5, ugan.py:163
    tsl = _ugan_decoder(sd, "tsl_decoder.", t_e5, t_sk, use_tanh=True)

    s_bot, s_sk = _ugan_encoder(sd, "seg_encoder.", x)
    s_e5 = basic_block(sd, "enc5.", s_bot)               # same module again, ugan.py:168
    seg = _ugan_decoder(sd, "seg_decoder.", s_e5, s_sk, use_tanh=False)

    if val_phase or not with_nce:
        return seg, tsl
    assert sample_ids is not None, "oracle: feed the captured patch ids"
    feats = [patch_sample(sd, t_e5, sample_ids[0])]
    return seg, tsl, feats, list(sample_ids)


# --------------------------------------------------------------------------- discriminator
def discriminator_forward(sd: SD, x):
    """network/ugan.py:198-229 -- returns (out_src [B,1,4,4], out_cls [B,n_modal])."""
    h = _act(F.conv2d(x, sd["main.0.weight"], sd["main.0.bias"], stride=2, padding=1))
    i = 2
    while f"main.{i}.conv1.weight" in sd:
        h = bottle_block(sd, f"main.{i}.", h, stride=2)
        i += 1
    src = F.conv2d(h, sd["conv_src.weight"], padding=1)
    cls = F.conv2d(h, sd["conv_cls.weight"])
    return src, cls.view(cls.size(0), cls.size(1))


def gradient_penalty(out_src, x_hat):
    """trainer/uganShp0Trainer.py:127-134."""
    dydx = torch.autograd.grad(out_src, x_hat, torch.ones_like(out_src),
                               retain_graph=True, create_graph=True)[0]
    n = torch.sqrt(torch.sum(dydx.reshape(dydx.size(0), -1) ** 2, dim=1))
    return torch.mean((n - 1) ** 2)


# --------------------------------------------------------------------------- losses
def soft_dice(logits, labels, batch_dice=True):
    """misc/loss.py:23-63."""
    prob = F.softmax(logits, dim=1)
    onehot = torch.zeros_like(prob).scatter_(1, labels.unsqueeze(1), 1.0)
    dims = (0, 2, 3) if batch_dice else (2, 3)
    tp = (prob * onehot).sum(dims)
    fp = (prob * (1 - onehot)).sum(dims)
    fn = ((1 - prob) * onehot).sum(dims)
    dc = (2 * tp + DICE_SMOOTH) / (2 * tp + fp + fn + DICE_SMOOTH + DICE_EPS)
    dc = dc[1:] if batch_dice else dc[:, 1:]
    return 1.0 - dc.mean()


def dice_ce(logits, labels, weight_ce=0.5, weight_dc=0.5, batch_dice=True):
    """misc/loss.py:8-20 with config.py:32-33 weights and baseTrainer.py:57 batch_dice=True."""
    return weight_dc * soft_dice(logits, labels, batch_dice) + weight_ce * F.cross_entropy(logits, labels)


def patch_nce(feat_q, feat_k, batch_size):
    """network/patchnce.py:13-51 -- per-row loss; ``batch_size`` is the ctor argument."""
    n, dim = feat_q.shape
    feat_k = feat_k.detach()
    l_pos = (feat_q * feat_k).sum(1, keepdim=True)
    q = feat_q.view(batch_size, -1, dim)
    k = feat_k.view(batch_size, -1, dim)
    npatch = q.size(1)
    l_neg = torch.bmm(q, k.transpose(2, 1))
    eye = torch.eye(npatch, dtype=torch.bool)[None]
    l_neg = l_neg.masked_fill(eye, -10.0).view(-1, npatch)
    out = torch.cat([l_pos, l_neg], dim=1) / NCE_T
    return F.cross_entropy(out, torch.zeros(n, dtype=torch.long), reduction="none")


def sigmoid_rampup(current, rampup_length):
    """trainer/baseTrainer.py:65-72."""
    if rampup_length == 0:
        return 1.0
    c = min(max(float(current), 0.0), float(rampup_length))
    ph = 1.0 - c / rampup_length
    return float(math.exp(-5.0 * ph * ph))


def poly_lr(base_lr, it, max_it):
    """trainer/uganConsisTrainer.py:198 / unetTrainer.py:80."""
    return base_lr * (1.0 - it / max_it) ** 0.9


def onehot(idx, dim):
    """trainer/uganShp0Trainer.py:109-113."""
    out = torch.zeros(idx.size(0), dim)
    out[torch.arange(idx.size(0)), idx.long()] = 1
    return out


def medpy_dc(result, reference):
    """medpy.metric.binary.dc restated from its published formula (parity unpinned):
    2|A and B| / (|A| + |B|), 0.0 when both are empty.  Used at misc/utils.py:192."""
    import numpy as np
    a = np.asarray(result).astype(bool)
    b = np.asarray(reference).astype(bool)
    inter = np.count_nonzero(a & b)
    sa, sb = np.count_nonzero(a), np.count_nonzero(b)
    try:
        return 2.0 * inter / float(sa + sb)
    except ZeroDivisionError:
        return 0.0


# --------------------------------------------------------------------------- trainer arithmetic
def unet_train_step(sd: SD, opt: torch.optim.Optimizer, img, msk, it: int,
                    base_lr=1e-2, max_it=30000):
    """trainer/unetTrainer.py:56-85: fwd, DiceCE, zero_grad, backward, SGD step, poly LR."""
    out = unet_forward(sd, img)
    loss = dice_ce(out, msk)
    opt.zero_grad()
    loss.backward()
    opt.step()
    lr_ = poly_lr(base_lr, it, max_it)
    for g in opt.param_groups:
        g["lr"] = lr_
    return float(loss.item()), out.detach()


def ugan_consis_iteration(g_sd: SD, d_sd: SD, g_opt, d_opt, x_real, y_real, modal_org, mj: int,
                          alpha, sample_ids, it: int, epoch: int, nce_batch: int,
                          n_modal=4, lambda_cls=1.0, lambda_rec=10.0, lambda_gp=10.0, lambda_seg=10.0,
                          lambda_semi_base=10.0, max_epoch=200, base_lr=1e-2, max_it=30000,
                          semi_start_iter=1000):
    """One iteration of trainer/uganConsisTrainer.py:110-203 (n_critic = 1).

    RNG-dependent quantities (``mj`` :114, ``alpha`` :138, ``sample_ids`` from randperm
    ugan.py:321-323) are inputs.  Returns the 10 logged scalars (:148-149,:183-188).
    """
    bs = y_real.size(0)                                   # cfg.batch_size = labeled half
    lambda_semi = lambda_semi_base * sigmoid_rampup(epoch, max_epoch)      # :74
    modal_trg = torch.full_like(modal_org, mj)
    vec_org, vec_trg = onehot(modal_org, n_modal), onehot(modal_trg, n_modal)
    vec_ot, vec_to = vec_trg - vec_org, vec_org - vec_trg

    # ---- D-step (:129-146)
    out_src, out_cls = discriminator_forward(d_sd, x_real)
    d_real = -out_src.mean()
    d_cls = F.cross_entropy(out_cls, modal_org)
    _, x_fake, _, _ = ugan_forward(g_sd, x_real, vec_ot, sample_ids, n_modal=n_modal)
    out_src, _ = discriminator_forward(d_sd, x_fake.detach())
    d_fake = out_src.mean()
    x_hat = (alpha * x_real.detach() + (1 - alpha) * x_fake.detach()).requires_grad_(True)
    out_src, _ = discriminator_forward(d_sd, x_hat)
    d_gp = gradient_penalty(out_src, x_hat)
    d_loss = d_real + d_fake + lambda_cls * d_cls + lambda_gp * d_gp
    d_opt.zero_grad(); g_opt.zero_grad()
    d_loss.backward()
    d_opt.step()
    logs = dict(D_real=d_real.item(), D_fake=d_fake.item(), D_cls=d_cls.item(), D_gp=d_gp.item())

    # ---- G-step (:150-180)
    y_fake, x_fake, feat_x, _ = ugan_forward(g_sd, x_real, vec_ot, sample_ids, n_modal=n_modal)
    out_src, out_cls = discriminator_forward(d_sd, x_fake)
    g_fake = -out_src.mean()
    g_cls = F.cross_entropy(out_cls, modal_trg)
    g_seg = dice_ce(y_fake[:bs], y_real)
    y_rec, x_rec, feat_f, _ = ugan_forward(g_sd, x_fake, vec_to, sample_ids, n_modal=n_modal)
    g_rec = (x_real - x_rec).abs().mean()
    if it < semi_start_iter:
        g_semi = torch.zeros(())
    else:
        g_semi = dice_ce(y_rec, y_fake.argmax(dim=1))      # :45-53
    g_nce = sum(patch_nce(ff, fx, nce_batch).mean() for ff, fx in zip(feat_f, feat_x)) / len(feat_f)  # :55-64
    g_loss = g_fake + lambda_rec * g_rec + lambda_cls * g_cls + lambda_seg * g_seg \
        + lambda_semi * g_semi + 1.0 * g_nce
    d_opt.zero_grad(); g_opt.zero_grad()
    g_loss.backward()
    g_opt.step()
    logs.update(G_fake=g_fake.item(), G_rec=g_rec.item(), G_cls=g_cls.item(), G_seg=g_seg.item(),
                G_semi=float(g_semi.item()), G_nce=g_nce.item())

    lr_ = poly_lr(base_lr, it, max_it)                     # :198-202
    for opt in (g_opt, d_opt):
        for grp in opt.param_groups:
            grp["lr"] = lr_
    return logs, dict(seg=y_fake.detach(), tsl=x_fake.detach())


# --------------------------------------------------------------------------- sibling trainers
def mean_teacher_iteration(sd: SD, ema_sd: SD, opt, img, msk, noise, it: int, epoch: int,
                           lambda_semi_base=1.0, epoch_rampup=30, ema_decay=0.99, base_lr=1e-2, max_it=30000,
                           semi_start_iter=100):
    """trainer/meanTeacherTrainer.py:86-149: student on [labeled | unlabeled], EMA teacher on the noised unlabeled
    half, DiceCE + rampup * mean((softmax - softmax)^2) (from iteration 100), SGD, EMA update (:63-69), poly LR.
    ``noise`` (:104, clamp(randn * 0.01)) is an input.  Returns (seg_loss, semi_loss)."""
    bs = msk.size(0)
    lambda_semi = lambda_semi_base * sigmoid_rampup(epoch, epoch_rampup)
    out = unet_forward(sd, img)
    with torch.no_grad():
        ema_soft = torch.softmax(unet_forward(ema_sd, img[bs:] + noise), dim=1)
    seg = dice_ce(out[:bs], msk)
    semi = torch.zeros(()) if it < semi_start_iter else torch.mean((torch.softmax(out, dim=1)[bs:] - ema_soft) ** 2)
    total = seg + lambda_semi * semi
    opt.zero_grad()
    total.backward()
    opt.step()
    alpha = 0.0 if it < 100 else min(1 - 1 / (it + 1), ema_decay)
    with torch.no_grad():
        for k in ema_sd:
            ema_sd[k].mul_(alpha).add_(sd[k].detach(), alpha=1 - alpha)
    lr_ = poly_lr(base_lr, it, max_it)
    for g in opt.param_groups:
        g["lr"] = lr_
    return float(seg.item()), float(semi.item())


def cross_pse_iteration(sd1: SD, sd2: SD, opt1, opt2, img, msk, it: int, epoch: int, lambda_semi_base=0.1,
                        max_epoch=200, base_lr=1e-2, max_it=30000):
    """trainer/crossPseTrainer.py:84-146: two U-Nets, DiceCE on the labeled half, each net's unlabeled half against
    the OTHER net's argmax.  Returns (seg1, seg2, semi1, semi2)."""
    bs = msk.size(0)
    lambda_semi = lambda_semi_base * sigmoid_rampup(epoch, max_epoch)
    out1 = unet_forward(sd1, img)
    s1 = dice_ce(out1[:bs], msk)
    out2 = unet_forward(sd2, img)
    s2 = dice_ce(out2[:bs], msk)
    pred1 = torch.argmax(out1[bs:], dim=1).detach()
    pred2 = torch.argmax(out2[bs:], dim=1).detach()
    semi1 = dice_ce(out1[bs:], pred2)
    semi2 = dice_ce(out2[bs:], pred1)
    total = s1 + s2 + lambda_semi * semi1 + lambda_semi * semi2
    opt1.zero_grad(); opt2.zero_grad()
    total.backward()
    opt1.step(); opt2.step()
    lr_ = poly_lr(base_lr, it, max_it)
    for opt in (opt1, opt2):
        for g in opt.param_groups:
            g["lr"] = lr_
    return tuple(float(t.item()) for t in (s1, s2, semi1, semi2))


def ugan_iteration(g_sd: SD, d_sd: SD, g_opt, d_opt, x_real, y_real, modal_org, mj: int, alpha, it: int, epoch: int,
                   n_modal=4, lambda_cls=1.0, lambda_rec=10.0, lambda_gp=10.0, lambda_seg=10.0, lambda_shp_base=10.0,
                   lambda_shp_lazy=20.0, base_lr=1e-2, max_it=30000):
    """trainer/uganTrainer.py:134-222 (UGAN without the NCE head, all slices labeled, shape term on the cycle's
    segmentation with an epoch-ramped weight :122-123).  Returns the 9 scalars D_real..G_shp."""
    lambda_shp = min(epoch * (lambda_shp_base / lambda_shp_lazy), lambda_seg)
    modal_trg = torch.full_like(modal_org, mj)
    vec_org, vec_trg = onehot(modal_org, n_modal), onehot(modal_trg, n_modal)
    vec_ot, vec_to = vec_trg - vec_org, vec_org - vec_trg
    out_src, out_cls = discriminator_forward(d_sd, x_real)
    d_real = -out_src.mean()
    d_cls = F.cross_entropy(out_cls, modal_org)
    _, x_fake = ugan_forward(g_sd, x_real, vec_ot, n_modal=n_modal, with_nce=False)
    out_src, _ = discriminator_forward(d_sd, x_fake.detach())
    d_fake = out_src.mean()
    x_hat = (alpha * x_real.detach() + (1 - alpha) * x_fake.detach()).requires_grad_(True)
    out_src, _ = discriminator_forward(d_sd, x_hat)
    d_gp = gradient_penalty(out_src, x_hat)
    d_loss = d_real + d_fake + lambda_cls * d_cls + lambda_gp * d_gp
    d_opt.zero_grad(); g_opt.zero_grad()
    d_loss.backward()
    d_opt.step()
    y_fake, x_fake = ugan_forward(g_sd, x_real, vec_ot, n_modal=n_modal, with_nce=False)
    out_src, out_cls = discriminator_forward(d_sd, x_fake)
    g_fake = -out_src.mean()
    g_cls = F.cross_entropy(out_cls, modal_trg)
    g_seg = dice_ce(y_fake, y_real)
    y_rec, x_rec = ugan_forward(g_sd, x_fake, vec_to, n_modal=n_modal, with_nce=False)
    g_rec = (x_real - x_rec).abs().mean()
    g_shp = dice_ce(y_rec, y_real)
    g_loss = g_fake + lambda_rec * g_rec + lambda_cls * g_cls + lambda_seg * g_seg + lambda_shp * g_shp
    d_opt.zero_grad(); g_opt.zero_grad()
    g_loss.backward()
    g_opt.step()
    lr_ = poly_lr(base_lr, it, max_it)
    for opt in (g_opt, d_opt):
        for grp in opt.param_groups:
            grp["lr"] = lr_
    return tuple(float(t.item()) for t in (d_real, d_fake, d_cls, d_gp, g_fake, g_rec, g_cls, g_seg, g_shp))

