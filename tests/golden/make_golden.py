#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE's own modules on CPU and dumps small .npz fixtures.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``network.{unet,ugan,patchnce,networks}`` and ``misc.loss`` from /root/reference
(read-only), fills them from the deterministic weight recipe in ``oracle/recipe.py`` through
``load_state_dict`` (after asserting the recipe's key/shape tables equal the reference
modules' own state_dict), replays the arithmetic of ``trainer/unetTrainer.py:56-85`` and
``trainer/uganConsisTrainer.py:110-203`` with those modules (the trainer classes themselves
need medpy/torchvision/tensorboard, absent here -- SURVEY.md 8c), and stores inputs, outputs,
loss scalars and selected gradients.  Fixtures hold data only -- no reference source text.
"""
import os
import random
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SMSUT_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# ugan.py:329 calls mlp.cuda() unconditionally: make it a no-op on this GPU-less box.
nn.Module.cuda = lambda self, *a, **k: self

from network.unet import UNet                      # noqa: E402  (reference)
from network.ugan import UGANnce, Discriminator    # noqa: E402  (reference)
from network.patchnce import PatchNCELoss          # noqa: E402  (reference)
import network.networks as ref_networks            # noqa: E402  (reference)
from misc.loss import DiceAndCrossEntropyLoss      # noqa: E402  (reference)

from oracle import recipe                          # noqa: E402  (ours)

torch.set_num_threads(8)


def load(module, shapes, seed):
    sd_ref = module.state_dict()
    assert list(sd_ref.keys()) == list(shapes.keys()), "recipe key table != reference state_dict"
    for k, v in sd_ref.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, tuple(v.shape), shapes[k])
    module.load_state_dict(recipe.fill(shapes, seed))
    return module


def npy(t):
    return t.detach().cpu().numpy().copy()     # copy: state_dict()/grad tensors alias live memory on CPU


def grad_summary(module, full_keys=()):
    out = {}
    names, norms = [], []
    for k, p in module.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k in full_keys:
            out["grad::" + k] = npy(p.grad)
    out["grad_names"] = np.array(names)
    out["grad_l2"] = np.array(norms, dtype=np.float64)
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------------------------
def gen_unet_small():
    """UNet(1,3,4,'instance','lrelu') on 2x1x64x64: logits, DiceCE, grads, 2 SGD steps (unetTrainer.py:56-85)."""
    seed, B, H, ncls, w = 11, 2, 64, 3, 4
    net = load(UNet(1, ncls, w, norm_type="instance", act_type="lrelu"), recipe.unet_shapes(1, ncls, w), seed)
    net.train()
    x = recipe.synth_images((B, 1, H, H), seed + 1)
    y = recipe.synth_labels(B, H, H, ncls, seed + 2, block=8)
    crit = DiceAndCrossEntropyLoss(weight_ce=0.5, weight_dc=0.5, batch_dice=True)
    opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    losses = []
    extra = {}
    for it in range(2):
        out = net(x)
        loss = crit(out, y)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            extra.update(logits=npy(out))
            extra.update(grad_summary(net, ("encoder.pre_conv.weight", "decoder.fc.weight",
                                            "encoder.layer1.bn1.weight", "encoder.layer1.shortcut1.weight",
                                            "decoder.up1.up.weight", "decoder.layer4.conv1.weight")))
        opt.step()
        lr_ = 1e-2 * (1.0 - it / 30000) ** 0.9
        for g in opt.param_groups:
            g["lr"] = lr_
        losses.append(loss.item())
    sd = net.state_dict()
    save("unet_small", seed=seed, B=B, H=H, ncls=ncls, w=w, x=npy(x), y=npy(y),
         losses=np.array(losses), post_pre_conv=npy(sd["encoder.pre_conv.weight"]),
         post_fc=npy(sd["decoder.fc.weight"]), **extra)


def gen_unet_relu():
    """UNet default act ('relu') with instance norm -- the constructor surface beyond lrelu."""
    seed, B, H, ncls, w = 17, 1, 32, 2, 4
    net = load(UNet(1, ncls, w, norm_type="instance", act_type="relu"), recipe.unet_shapes(1, ncls, w), seed)
    x = recipe.synth_images((B, 1, H, H), seed + 1)
    save("unet_relu", seed=seed, B=B, H=H, ncls=ncls, w=w, x=npy(x), logits=npy(net(x)))


def gen_unet_256():
    """Full-size UNet(1,5,16) on one 256x256 slice: strided logits + loss + per-class sums."""
    seed, B, H, ncls, w = 23, 1, 256, 5, 16
    net = load(UNet(1, ncls, w, norm_type="instance", act_type="lrelu"), recipe.unet_shapes(1, ncls, w), seed)
    x = recipe.synth_images((B, 1, H, H), seed + 1)
    y = recipe.synth_labels(B, H, H, ncls, seed + 2)
    out = net(x)
    loss = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)(out, y)
    loss.backward()
    save("unet_256", seed=seed, B=B, H=H, ncls=ncls, w=w, logits_s8=npy(out[:, :, ::8, ::8]),
         logits_sum=npy(out.double().sum((0, 2, 3))), logits_abs_sum=npy(out.double().abs().sum((0, 2, 3))),
         loss=loss.item(), **grad_summary(net, ("decoder.fc.weight",)))


def gen_disc_small():
    """Discriminator(64,4,4,max_width=32) on 3x1x64x64 incl. WGAN-GP double backward
    (uganShp0Trainer.py:127-134) and the D-loss weight grads."""
    seed, B, S, nm, w, mw = 31, 3, 64, 4, 4, 32
    D = load(Discriminator(S, nm, w, max_width=mw), recipe.disc_shapes(S, nm, w, mw), seed)
    D.train()
    x = recipe.synth_images((B, 1, S, S), seed + 1)
    xf = recipe.synth_images((B, 1, S, S), seed + 2)
    alpha = torch.from_numpy(np.random.RandomState(seed + 3).standard_normal((B, 1, 1, 1))).float()
    modal = torch.tensor([0, 2, 3])
    src, cls = D(x)
    d_real = -src.mean()
    d_cls = F.cross_entropy(cls, modal)
    src_f, _ = D(xf)
    d_fake = src_f.mean()
    x_hat = (alpha * x + (1 - alpha) * xf).requires_grad_(True)
    src_h, _ = D(x_hat)
    dydx = torch.autograd.grad(src_h, x_hat, torch.ones_like(src_h), retain_graph=True, create_graph=True)[0]
    gp = torch.mean((torch.sqrt(torch.sum(dydx.view(B, -1) ** 2, dim=1)) - 1) ** 2)
    d_loss = d_real + d_fake + 1.0 * d_cls + 10.0 * gp
    D.zero_grad()
    d_loss.backward()
    save("disc_small", seed=seed, B=B, S=S, nm=nm, w=w, mw=mw, x=npy(x), xf=npy(xf), alpha=npy(alpha),
         modal=npy(modal), out_src=npy(src), out_cls=npy(cls), dydx=npy(dydx),
         scalars=np.array([d_real.item(), d_fake.item(), d_cls.item(), gp.item()]),
         **grad_summary(D, ("main.0.weight", "main.0.bias", "main.2.bn1.weight", "main.3.downsample.0.weight",
                            "conv_src.weight", "conv_cls.weight", "main.4.conv2.weight")))


def gen_ugan_small():
    """UGANnce(1,5,4,16) on 4x1x64x64 (netF needs base_width 16): seg, tsl, feats."""
    seed, B, H = 41, 4, 64
    G = load(UGANnce(1, 5, 4, 16), recipe.ugan_shapes(1, 5, 4, 16), seed)
    G.train()
    x = recipe.synth_images((B, 1, H, H), seed + 1)
    m = torch.tensor([[-1., 0, 1, 0]] * 2 + [[0., 0, 0, 0]] + [[0., -1, 0, 1]])
    ids = torch.from_numpy(np.random.RandomState(seed + 2).permutation(16)[:64].astype(np.int64))
    seg, tsl, feats, rid = G(x, m, sample_ids=[ids])
    seg_v, tsl_v = G(x, val_phase=True)
    save("ugan_small", seed=seed, B=B, H=H, x=npy(x), m=npy(m), ids=npy(ids), seg=npy(seg), tsl=npy(tsl),
         feat=npy(feats[0]), seg_val=npy(seg_v), tsl_val=npy(tsl_v))


def gen_losses():
    """misc/loss.py, network/patchnce.py, networks.Normalize on random tensors."""
    rs = np.random.RandomState(53)
    logits = torch.from_numpy(rs.standard_normal((3, 5, 16, 16)) * 2).float()
    labels = torch.from_numpy(rs.randint(0, 5, size=(3, 16, 16)).astype(np.int64))
    l_bd = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)(logits, labels).item()
    l_sd = DiceAndCrossEntropyLoss(1.0, 1.0, batch_dice=False)(logits, labels).item()
    q = torch.from_numpy(rs.standard_normal((64, 32))).float()
    k = torch.from_numpy(rs.standard_normal((64, 32))).float()
    qn, kn = ref_networks.Normalize(2)(q), ref_networks.Normalize(2)(k)
    nce = PatchNCELoss(2)(qn, kn)
    save("losses", logits=npy(logits), labels=npy(labels), dicece_batch=l_bd, dicece_sample=l_sd,
         q=npy(q), k=npy(k), qn=npy(qn), nce=npy(nce))


def _seeded_fill(module, seed):
    """Deterministic values for modules whose key table is taken from the reference itself (networks.py zoo)."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith("filt"):
            new[k] = v.clone()
            continue
        rs = np.random.RandomState((zlib_crc(k) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        a = rs.standard_normal(tuple(v.shape))
        a = a / np.sqrt(np.prod(v.shape[1:])) if v.dim() >= 2 else 0.1 * a
        new[k] = torch.from_numpy(np.ascontiguousarray(a)).float()
    module.load_state_dict(new)
    return new


def zlib_crc(k):
    import zlib
    return zlib.crc32(k.encode())


def gen_networks_zoo():
    """networks.ResnetGenerator / NLayerDiscriminator / PatchDiscriminator / Downsample / Upsample (SURVEY 8a rows
    13-14): small instances with get_norm_layer('instance'); key/shape tables + outputs + a few grads."""
    norm = ref_networks.get_norm_layer("instance")
    seed = 71
    G = ref_networks.ResnetGenerator(1, 1, ngf=8, norm_layer=norm, n_blocks=2)
    w = _seeded_fill(G, seed)
    x = recipe.synth_images((2, 1, 32, 32), seed + 1).requires_grad_(True)
    y = G(x)
    gy = torch.from_numpy(np.random.RandomState(seed + 2).standard_normal(tuple(y.shape))).float()
    y.backward(gy)
    feats = G(x.detach(), layers=[0, 4, 8], encode_only=True)
    rec = dict(seed=seed, g_keys=np.array(list(w.keys())), g_shapes=np.array([str(tuple(v.shape)) for v in w.values()]),
               g_x=npy(x), g_y=npy(y), g_gy=npy(gy), g_gx=npy(x.grad), g_feat8=npy(feats[2]),
               **{"g_" + k: v for k, v in grad_summary(G, ("model.1.weight", "model.12.conv_block.1.bias", "model.19.bias")).items()})
    D = ref_networks.NLayerDiscriminator(1, ndf=8, n_layers=3, norm_layer=norm)
    wd = _seeded_fill(D, seed + 5)
    xd = recipe.synth_images((2, 1, 64, 64), seed + 6).requires_grad_(True)
    yd = D(xd)
    gyd = torch.from_numpy(np.random.RandomState(seed + 7).standard_normal(tuple(yd.shape))).float()
    yd.backward(gyd)
    rec.update(d_keys=np.array(list(wd.keys())), d_shapes=np.array([str(tuple(v.shape)) for v in wd.values()]),
               d_x=npy(xd), d_y=npy(yd), d_gy=npy(gyd), d_gx=npy(xd.grad),
               **{"d_" + k: v for k, v in grad_summary(D, ("model.0.weight", "model.3.weight", "model.11.bias")).items()})
    P = ref_networks.PatchDiscriminator(1, ndf=8, norm_layer=norm)
    _seeded_fill(P, seed + 9)
    rec["p_y"] = npy(P(recipe.synth_images((1, 1, 32, 32), seed + 10)))
    t = recipe.synth_images((2, 3, 10, 12), seed + 11)
    rec.update(t=npy(t), down=npy(ref_networks.Downsample(3)(t)), up=npy(ref_networks.Upsample(3)(t)))
    save("networks_zoo", **rec)


def gen_iter_small(dtype=torch.float32, name="iter_small"):
    """Two uganConsis iterations (uganConsisTrainer.py:110-203) at 64x64, B = 2 labeled + 2 unlabeled,
    PatchNCELoss(2) fed B=4 (the reference's batch_size/B mismatch), iter >= 1000 so the
    consistency branch runs; all 10 scalars per iteration + post-step weight slices."""
    seed, bs, H, nm = 61, 2, 64, 4
    B = 2 * bs
    torch.set_default_dtype(dtype)
    G = load(UGANnce(1, 5, nm, 16), recipe.ugan_shapes(1, 5, nm, 16), seed).to(dtype)
    D = load(Discriminator(H, nm, 16, max_width=256), recipe.disc_shapes(H, nm, 16, 256), seed + 1).to(dtype)
    G.train(); D.train()
    crit = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)
    nce = PatchNCELoss(bs)
    g_opt = torch.optim.SGD(G.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    d_opt = torch.optim.Adam(D.parameters(), 1e-2, [0.9, 0.999], weight_decay=1e-3)
    lam = dict(cls=1.0, rec=10.0, gp=10.0, seg=10.0)
    epoch, it0 = 100, 15000
    ph = 1.0 - epoch / 200.0
    lam_semi = 10.0 * float(np.exp(-5.0 * ph * ph))

    def onehot(idx):
        o = torch.zeros(idx.size(0), nm)
        o[np.arange(idx.size(0)), idx.long()] = 1
        return o

    rec = dict(seed=seed, bs=bs, H=H, nm=nm, epoch=epoch, it0=it0)
    logs_all = []
    for step in range(2):
        it = it0 + step
        x_real = recipe.synth_images((B, 1, H, H), seed + 10 + step).to(dtype)
        y_real = recipe.synth_labels(bs, H, H, 5, seed + 20 + step, block=8)
        modal_org = torch.tensor([1] * bs + [3] * bs)
        mj = (step + 2) % nm
        alpha = torch.from_numpy(np.random.RandomState(seed + 30 + step).standard_normal((B, 1, 1, 1))).float().to(dtype)
        ids = torch.from_numpy(np.random.RandomState(seed + 40 + step).permutation(16)[:64].astype(np.int64))
        modal_trg = torch.zeros_like(modal_org).fill_(mj)
        vec_org, vec_trg = onehot(modal_org), onehot(modal_trg)
        vec_ot, vec_to = vec_trg - vec_org, vec_org - vec_trg

        out_src, out_cls = D(x_real)
        d_real = -torch.mean(out_src)
        d_cls = F.cross_entropy(out_cls, modal_org)
        _, x_fake, _, _ = G(x_real, vec_ot, sample_ids=[ids])
        out_src, out_cls = D(x_fake.detach())
        d_fake = torch.mean(out_src)
        x_hat = (alpha * x_real.data + (1 - alpha) * x_fake.data).requires_grad_(True)
        out_src, _ = D(x_hat)
        dydx = torch.autograd.grad(out_src, x_hat, torch.ones(out_src.size()), retain_graph=True,
                                   create_graph=True, only_inputs=True)[0]
        d_gp = torch.mean((torch.sqrt(torch.sum(dydx.view(B, -1) ** 2, dim=1)) - 1) ** 2)
        d_loss = d_real + d_fake + lam["cls"] * d_cls + lam["gp"] * d_gp
        d_opt.zero_grad(); g_opt.zero_grad()
        d_loss.backward()
        if step == 0:
            for k, v in grad_summary(D, ("conv_cls.weight", "main.0.weight")).items():
                rec["D0_" + k] = v
        d_opt.step()
        if step == 0:      # Adam's first update is +-lr per element: keep the post-step-0 weights to check it element-wise
            rec["post0_D_cls"] = npy(D.state_dict()["conv_cls.weight"])
            rec["post0_D_stem"] = npy(D.state_dict()["main.0.weight"])
            rec["post0_D_bn"] = npy(D.state_dict()["main.2.bn1.weight"])

        y_fake, x_fake, feat_x, _ = G(x_real, vec_ot, sample_ids=[ids])
        out_src, out_cls = D(x_fake)
        g_fake = -torch.mean(out_src)
        g_cls = F.cross_entropy(out_cls, modal_trg)
        g_seg = crit(y_fake[:bs], y_real)
        y_rec, x_rec, feat_f, _ = G(x_fake, vec_to, sample_ids=[ids])
        g_rec = torch.mean(torch.abs(x_real - x_rec))
        g_semi = crit(y_rec, torch.argmax(y_fake, dim=1))
        g_nce = sum((nce(ff, fx) * 1.0).mean() for ff, fx in zip(feat_f, feat_x)) / 1
        g_loss = g_fake + lam["rec"] * g_rec + lam["cls"] * g_cls + lam["seg"] * g_seg + lam_semi * g_semi + g_nce
        d_opt.zero_grad(); g_opt.zero_grad()
        g_loss.backward()
        if step == 0:
            for k, v in grad_summary(G, ("seg_decoder.fc.weight", "tsl_decoder.fc.bias", "netF.mlp_0.2.bias",
                                         "tsl_encoder.pre.0.weight")).items():
                rec["G0_" + k] = v
            rec["seg0_s4"] = npy(y_fake[:, :, ::4, ::4])
            rec["tsl0"] = npy(x_fake)
        g_opt.step()
        if step == 0:
            rec["post0_G_seg_fc"] = npy(G.state_dict()["seg_decoder.fc.weight"])
            rec["post0_G_tsl_pre"] = npy(G.state_dict()["tsl_encoder.pre.0.weight"])
        lr_ = 1e-2 * (1.0 - it / 30000) ** 0.9
        for grp in list(g_opt.param_groups) + list(d_opt.param_groups):
            grp["lr"] = lr_
        logs_all.append([d_real.item(), d_fake.item(), d_cls.item(), d_gp.item(), g_fake.item(), g_rec.item(),
                         g_cls.item(), g_seg.item(), g_semi.item(), g_nce.item()])
        rec[f"mj{step}"] = mj
    rec["scalar_names"] = np.array(["D_real", "D_fake", "D_cls", "D_gp", "G_fake", "G_rec", "G_cls", "G_seg",
                                    "G_semi", "G_nce"])
    rec["scalars"] = np.array(logs_all, dtype=np.float64)
    gsd, dsd = G.state_dict(), D.state_dict()
    rec["post_G_seg_fc"] = npy(gsd["seg_decoder.fc.weight"])
    rec["post_G_tsl_pre"] = npy(gsd["tsl_encoder.pre.0.weight"])
    rec["post_D_cls"] = npy(dsd["conv_cls.weight"])
    rec["post_D_stem"] = npy(dsd["main.0.weight"])
    save(name, **rec)
    torch.set_default_dtype(torch.float32)


def gen_iter_small_f64():
    """The same two iterations with the reference's modules in fp64: the spread against ``iter_small`` is the reference's own
    sensitivity to rounding, which bounds the bands of tests/test_trainer_gpu.py (tests/trace_bands.py)."""
    gen_iter_small(dtype=torch.float64, name="iter_small_f64")


def gen_siblings():
    """One/two iterations of the sibling trainers' arithmetic with the reference modules:
    meanTeacherTrainer.py:86-149, crossPseTrainer.py:84-146, uganTrainer.py:134-222."""
    from network.ugan import UGAN
    H, bs = 64, 2
    crit = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)
    rec = dict(H=H, bs=bs)

    def rampup(cur, length):
        ph = 1.0 - np.clip(cur, 0.0, length) / length
        return float(np.exp(-5.0 * ph * ph))

    # ---- mean teacher: UNet(1,3,8) student + teacher, iterations 150 and 151 (consistency on, EMA alpha > 0)
    stu = load(UNet(1, 3, 8, "instance", "lrelu"), recipe.unet_shapes(1, 3, 8), 71)
    ema = load(UNet(1, 3, 8, "instance", "lrelu"), recipe.unet_shapes(1, 3, 8), 72)
    for p in ema.parameters():
        p.detach_()
    stu.train(); ema.train()
    opt = torch.optim.SGD(stu.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    mt = []
    epoch = 20
    for step in range(2):
        it = 150 + step
        img = recipe.synth_images((2 * bs, 1, H, H), 73 + step)
        msk = recipe.synth_labels(bs, H, H, 3, 75 + step, block=8)
        noise = torch.clamp(torch.from_numpy(np.random.RandomState(77 + step).standard_normal((bs, 1, H, H))).float() * 0.01,
                            -0.02, 0.02)
        out = stu(img)
        out_soft = torch.softmax(out, dim=1)
        with torch.no_grad():
            ema_soft = torch.softmax(ema(img[bs:] + noise), dim=1)
        seg = crit(out[:bs], msk)
        semi = torch.mean((out_soft[bs:] - ema_soft) ** 2)
        total = seg + 1 * rampup(epoch, 30) * semi
        opt.zero_grad(); total.backward(); opt.step()
        alpha = min(1 - 1 / (it + 1), 0.99)
        for ep, p_ in zip(ema.parameters(), stu.parameters()):
            ep.data.mul_(alpha).add_(p_.data, alpha=1 - alpha)
        for g in opt.param_groups:
            g["lr"] = 1e-2 * (1.0 - it / 30000) ** 0.9
        mt.append([seg.item(), semi.item()])
    rec["mt_scalars"] = np.array(mt, dtype=np.float64)
    rec["mt_epoch"] = epoch
    rec["mt_post_fc"] = npy(stu.state_dict()["decoder.fc.weight"])
    rec["mt_post_ema_fc"] = npy(ema.state_dict()["decoder.fc.weight"])
    rec["mt_post_ema_pre"] = npy(ema.state_dict()["encoder.pre_conv.weight"])

    # ---- cross pseudo supervision: two UNet(1,3,8), one iteration
    n1 = load(UNet(1, 3, 8, "instance", "lrelu"), recipe.unet_shapes(1, 3, 8), 81)
    n2 = load(UNet(1, 3, 8, "instance", "lrelu"), recipe.unet_shapes(1, 3, 8), 82)
    n1.train(); n2.train()
    o1 = torch.optim.SGD(n1.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    o2 = torch.optim.SGD(n2.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    img = recipe.synth_images((2 * bs, 1, H, H), 83)
    msk = recipe.synth_labels(bs, H, H, 3, 84, block=8)
    epoch = 100
    out1 = n1(img); s1 = crit(out1[:bs], msk)
    out2 = n2(img); s2 = crit(out2[:bs], msk)
    pred1 = torch.argmax(out1[bs:], dim=1).detach(); pred2 = torch.argmax(out2[bs:], dim=1).detach()
    semi1 = crit(out1[bs:], pred2); semi2 = crit(out2[bs:], pred1)
    lam = 0.1 * rampup(epoch, 200)
    total = s1 + s2 + lam * semi1 + lam * semi2
    o1.zero_grad(); o2.zero_grad(); total.backward(); o1.step(); o2.step()
    rec["cp_scalars"] = np.array([s1.item(), s2.item(), semi1.item(), semi2.item()], dtype=np.float64)
    rec["cp_epoch"] = epoch
    rec["cp_post_fc1"] = npy(n1.state_dict()["decoder.fc.weight"])
    rec["cp_post_fc2"] = npy(n2.state_dict()["decoder.fc.weight"])
    rec["cp_pred1"] = npy(pred1).astype(np.uint8)

    # ---- UGANTrainer: UGAN(1,3,4,8) + Discriminator(64,4,8), one iteration at epoch 10 (lambda_shp = 5)
    nm = 4
    G = load(UGAN(1, 3, nm, 8), recipe.ugan_shapes(1, 3, nm, 8, nce=False), 91)
    D = load(Discriminator(H, nm, 8, max_width=512), recipe.disc_shapes(H, nm, 8, 512), 92)
    G.train(); D.train()
    g_opt = torch.optim.SGD(G.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    d_opt = torch.optim.Adam(D.parameters(), 1e-2, [0.9, 0.999], weight_decay=1e-3)
    B, epoch, it, mj = 2, 10, 1500, 2
    x_real = recipe.synth_images((B, 1, H, H), 93)
    y_real = recipe.synth_labels(B, H, H, 3, 94, block=8)
    modal_org = torch.tensor([1, 1])
    alpha = torch.from_numpy(np.random.RandomState(95).standard_normal((B, 1, 1, 1))).float()

    def onehot(idx):
        o = torch.zeros(idx.size(0), nm)
        o[np.arange(idx.size(0)), idx.long()] = 1
        return o
    modal_trg = torch.zeros_like(modal_org).fill_(mj)
    vec_org, vec_trg = onehot(modal_org), onehot(modal_trg)
    vec_ot, vec_to = vec_trg - vec_org, vec_org - vec_trg
    out_src, out_cls = D(x_real)
    d_real = -torch.mean(out_src); d_cls = F.cross_entropy(out_cls, modal_org)
    _, x_fake = G(x_real, vec_ot)
    out_src, out_cls = D(x_fake.detach()); d_fake = torch.mean(out_src)
    x_hat = (alpha * x_real.data + (1 - alpha) * x_fake.data).requires_grad_(True)
    out_src, _ = D(x_hat)
    dydx = torch.autograd.grad(out_src, x_hat, torch.ones(out_src.size()), retain_graph=True, create_graph=True,
                               only_inputs=True)[0]
    d_gp = torch.mean((torch.sqrt(torch.sum(dydx.view(B, -1) ** 2, dim=1)) - 1) ** 2)
    d_loss = d_real + d_fake + 1 * d_cls + 10 * d_gp
    d_opt.zero_grad(); g_opt.zero_grad(); d_loss.backward(); d_opt.step()
    y_fake, x_fake = G(x_real, vec_ot)
    out_src, out_cls = D(x_fake)
    g_fake = -torch.mean(out_src); g_cls = F.cross_entropy(out_cls, modal_trg)
    g_seg = crit(y_fake, y_real)
    y_rec, x_rec = G(x_fake, vec_to)
    g_rec = torch.mean(torch.abs(x_real - x_rec)); g_shp = crit(y_rec, y_real)
    lam_shp = min(epoch * (10 / 20), 10)
    g_loss = g_fake + 10 * g_rec + 1 * g_cls + 10 * g_seg + lam_shp * g_shp
    d_opt.zero_grad(); g_opt.zero_grad(); g_loss.backward()
    rec["ug_grad_seg_fc"] = npy(G.seg_decoder.fc.weight.grad)
    g_opt.step()
    rec["ug_scalars"] = np.array([d_real.item(), d_fake.item(), d_cls.item(), d_gp.item(), g_fake.item(), g_rec.item(),
                                  g_cls.item(), g_seg.item(), g_shp.item()], dtype=np.float64)
    rec["ug_epoch"], rec["ug_it"], rec["ug_mj"] = epoch, it, mj
    rec["ug_post_seg_fc"] = npy(G.state_dict()["seg_decoder.fc.weight"])
    rec["ug_tsl"] = npy(x_fake)
    save("siblings", **rec)


def gen_iter_trace(n_steps=None, dtype=torch.float32, name="iter_trace"):
    """A 32-iteration trajectory of the hot loop (uganConsisTrainer.py:110-203) at the BASELINE config-3 size:
    8 labeled + 8 unlabeled 256x256 slices, UGANnce(1,5,4,16) + Discriminator(256,4,16,256), PatchNCELoss(8) fed 16,
    iter >= 1000 (consistency on), SGD / Adam as uganShp0Trainer.py:72-74, poly LR.  Inputs and draws per step come from
    ``recipe.trace_inputs``; only the 10 scalars per step and a few weight norms are stored.  ``dtype=float64`` (name
    ``iter_trace_f64``) is the same replay in double precision: the spread between the two is the reference's OWN
    sensitivity to rounding, which is what bounds how closely any fp32 implementation can track the trace."""
    n_steps = int(os.environ.get("SMSUT_TRACE_STEPS", n_steps or 32))
    bs, H, nm = 8, 256, 4
    B = 2 * bs
    torch.set_default_dtype(dtype)
    G = load(UGANnce(1, 5, nm, 16), recipe.ugan_shapes(1, 5, nm, 16), 2020).to(dtype)
    D = load(Discriminator(H, nm, 16, max_width=256), recipe.disc_shapes(H, nm, 16, 256), 2021).to(dtype)
    G.train(); D.train()
    crit = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)
    nce = PatchNCELoss(bs)
    g_opt = torch.optim.SGD(G.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    d_opt = torch.optim.Adam(D.parameters(), 1e-2, [0.9, 0.999], weight_decay=1e-3)
    epoch, it0 = 100, 1000
    ph = 1.0 - epoch / 200.0
    lam_semi = 10.0 * float(np.exp(-5.0 * ph * ph))

    def onehot(idx):
        o = torch.zeros(idx.size(0), nm)
        o[np.arange(idx.size(0)), idx.long()] = 1
        return o

    import time
    logs_all, norms = [], []
    t0 = time.time()
    for step in range(n_steps):
        it = it0 + step
        x_real, y_real, modal_org, mj, alpha, ids = recipe.trace_inputs(step)
        x_real, alpha = x_real.to(dtype), alpha.to(dtype)
        modal_trg = torch.zeros_like(modal_org).fill_(mj)
        vec_org, vec_trg = onehot(modal_org), onehot(modal_trg)
        vec_ot, vec_to = vec_trg - vec_org, vec_org - vec_trg

        out_src, out_cls = D(x_real)
        d_real = -torch.mean(out_src)
        d_cls = F.cross_entropy(out_cls, modal_org)
        _, x_fake, _, _ = G(x_real, vec_ot, sample_ids=[ids])
        out_src, out_cls = D(x_fake.detach())
        d_fake = torch.mean(out_src)
        x_hat = (alpha * x_real.data + (1 - alpha) * x_fake.data).requires_grad_(True)
        out_src, _ = D(x_hat)
        dydx = torch.autograd.grad(out_src, x_hat, torch.ones(out_src.size()), retain_graph=True,
                                   create_graph=True, only_inputs=True)[0]
        d_gp = torch.mean((torch.sqrt(torch.sum(dydx.view(B, -1) ** 2, dim=1)) - 1) ** 2)
        d_loss = d_real + d_fake + 1.0 * d_cls + 10.0 * d_gp
        d_opt.zero_grad(); g_opt.zero_grad()
        d_loss.backward()
        d_opt.step()

        y_fake, x_fake, feat_x, _ = G(x_real, vec_ot, sample_ids=[ids])
        out_src, out_cls = D(x_fake)
        g_fake = -torch.mean(out_src)
        g_cls = F.cross_entropy(out_cls, modal_trg)
        g_seg = crit(y_fake[:bs], y_real)
        y_rec, x_rec, feat_f, _ = G(x_fake, vec_to, sample_ids=[ids])
        g_rec = torch.mean(torch.abs(x_real - x_rec))
        g_semi = crit(y_rec, torch.argmax(y_fake, dim=1))
        g_nce = sum((nce(ff, fx) * 1.0).mean() for ff, fx in zip(feat_f, feat_x)) / 1
        g_loss = g_fake + 10.0 * g_rec + 1.0 * g_cls + 10.0 * g_seg + lam_semi * g_semi + g_nce
        d_opt.zero_grad(); g_opt.zero_grad()
        g_loss.backward()
        gn = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in G.parameters() if p.grad is not None)))
        g_opt.step()
        lr_ = 1e-2 * (1.0 - it / 30000) ** 0.9
        for grp in list(g_opt.param_groups) + list(d_opt.param_groups):
            grp["lr"] = lr_
        logs_all.append([d_real.item(), d_fake.item(), d_cls.item(), d_gp.item(), g_fake.item(), g_rec.item(),
                         g_cls.item(), g_seg.item(), g_semi.item(), g_nce.item()])
        norms.append([float(torch.sqrt(sum(p.double().pow(2).sum() for p in G.parameters()))),
                      float(torch.sqrt(sum(p.double().pow(2).sum() for p in D.parameters()))), gn,
                      float(x_fake.abs().max())])
        print(f"trace[{name}] step {step}: " + " ".join(f"{v:.4g}" for v in logs_all[-1]) +
              f" | |G| {norms[-1][0]:.3f} |D| {norms[-1][1]:.3f} |gG| {gn:.3g}  ({time.time() - t0:.0f} s)", flush=True)
        # written after every step: a partial trace is still a usable fixture if the run is cut short
        save(name, scalars=np.array(logs_all, dtype=np.float64), norms=np.array(norms, dtype=np.float64),
             norm_names=np.array(["G_param_l2", "D_param_l2", "G_grad_l2", "x_fake_absmax"]),
             scalar_names=np.array(["D_real", "D_fake", "D_cls", "D_gp", "G_fake", "G_rec", "G_cls", "G_seg",
                                    "G_semi", "G_nce"]), it0=it0, epoch=epoch, bs=bs, H=H, g_seed=2020, d_seed=2021)
    torch.set_default_dtype(torch.float32)


def gen_iter_trace_f64():
    gen_iter_trace(dtype=torch.float64, name="iter_trace_f64")


def gen_validate():
    """The padded validation pass, replayed line by line with the reference's modules (trainer/uganShp0Trainer.py:250-287:
    eval mode, last batch zero-padded to cfg.batch_size, ``val_phase=True`` forward, crop, DiceCE per batch, argmax, volume
    assembly by name 'm_pid_z', meter accumulation with the PADDED batch size :270).  Stores the predicted volumes, the
    per-batch losses and the meter sums; the Dice matrix on top is medpy's ``dc`` (not installed: parity unpinned there)."""
    bs, H = 4, 64
    G = load(UGANnce(1, 5, 4, 16), recipe.ugan_shapes(1, 5, 4, 16), 77)
    G.eval()
    crit = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)
    batches = recipe.validation_batches(bs, H)
    gt = {}
    for _, msk, _, names in batches:
        for i, nm in enumerate(names):
            m, pid, z = nm.split("_")
            gt.setdefault(f"{m}_{pid}", {})[int(z)] = msk[i].numpy()
    gt = {k: np.stack([v[z] for z in sorted(v)]) for k, v in gt.items()}
    prd = {k: np.zeros(v.shape, dtype=v.dtype) for k, v in gt.items()}
    losses, n_prd, meter_sum, meter_n = [], 0, {}, {}
    logits_keep = None
    with torch.no_grad():
        for x_real, y_real, mdl, inm in batches:
            b, c, h, w = x_real.shape
            if b != bs:
                x_real = torch.cat([x_real, torch.zeros((bs - b, c, h, w), dtype=x_real.dtype)], dim=0)
            m = mdl[0].item()
            y_fake, x_fake = G(x_real, val_phase=True)
            if b != bs:
                y_fake, x_fake = y_fake[:b], x_fake[:b]
            sample_loss = crit(y_fake, y_real)
            losses.append(sample_loss.item())
            meter_sum[m] = meter_sum.get(m, 0.0) + sample_loss.item() * x_real.size(0)      # :270 -- padded size
            meter_n[m] = meter_n.get(m, 0) + x_real.size(0)
            pred = torch.argmax(y_fake, dim=1).numpy()
            if logits_keep is None:
                logits_keep = npy(y_fake[:, :, ::4, ::4])
            for i in range(b):
                mm, pid, z = inm[i].split("_")
                prd[f"{mm}_{pid}"][int(z)] = pred[i]
                n_prd += 1
    rec = dict(bs=bs, H=H, g_seed=77, n_prd=n_prd, losses=np.array(losses), logits0_s4=logits_keep,
               meter_keys=np.array(sorted(meter_sum)), meter_sum=np.array([meter_sum[k] for k in sorted(meter_sum)]),
               meter_n=np.array([meter_n[k] for k in sorted(meter_sum)]))
    for k in gt:
        rec["prd::" + k] = prd[k].astype(np.uint8)
        rec["gt::" + k] = gt[k].astype(np.uint8)
    save("validate", **rec)


def gen_augment_pil():
    """Geometry fixtures for the device-side joint augmentation: the reference's JointRotate / JointRandomResizedCrop
    (data_loader/externalTransforms.py:45-66) end in ``torchvision.transforms.functional.rotate`` / ``resized_crop`` on PIL images,
    i.e. ``Image.rotate(angle, resample)`` and ``Image.crop(box).resize(size, resample)`` (torchvision is not installed here; the two
    PIL calls it forwards to are the anchor) -- bilinear for the image, nearest for the mask.  Stored: one smooth 8-bit image and
    one block label map, PIL's results for three angles, three crop windows and the reference's ORDER of the two (rotate, then
    crop + resize: externalTransforms.py / baseLoader.py:15-85 compose them in that order).  JointElasticDeform (:69-90) needs
    ``elasticdeform`` (absent): not generated."""
    from PIL import Image
    H = W = 128
    rs = np.random.RandomState(5)
    coarse = rs.rand(H // 16 + 2, W // 16 + 2)
    img8 = np.clip(np.asarray(Image.fromarray((coarse * 255).astype(np.uint8)).resize((W, H), Image.BICUBIC), dtype=np.float32), 0, 255).astype(np.uint8)
    lab8 = np.repeat(np.repeat(rs.randint(0, 5, size=(H // 16, W // 16)).astype(np.uint8), 16, axis=0), 16, axis=1)
    angles = np.array([-15.0, 7.3, 12.0], dtype=np.float32)
    crops = np.array([[10, 18, 90, 100], [0, 0, 128, 128], [31, 2, 77, 115]], dtype=np.int64)     # (i, j, h, w)
    rec = dict(img=img8, lab=lab8, angles=angles, crops=crops)
    pi, pl = Image.fromarray(img8), Image.fromarray(lab8)
    for k, a in enumerate(angles):
        rec[f"rot_img_{k}"] = np.asarray(pi.rotate(float(a), Image.BILINEAR))
        rec[f"rot_lab_{k}"] = np.asarray(pl.rotate(float(a), Image.NEAREST))
    for k, (i, j, h, w) in enumerate(crops):
        rec[f"crop_img_{k}"] = np.asarray(pi.crop((j, i, j + w, i + h)).resize((W, H), Image.BILINEAR))
        rec[f"crop_lab_{k}"] = np.asarray(pl.crop((j, i, j + w, i + h)).resize((W, H), Image.NEAREST))
    for k in range(3):                                   # the composition, in the reference's order
        a = float(angles[k]); i, j, h, w = (int(v) for v in crops[(k + 1) % 3])
        rec[f"both_img_{k}"] = np.asarray(pi.rotate(a, Image.BILINEAR).crop((j, i, j + w, i + h)).resize((W, H), Image.BILINEAR))
        rec[f"both_lab_{k}"] = np.asarray(pl.rotate(a, Image.NEAREST).crop((j, i, j + w, i + h)).resize((W, H), Image.NEAREST))
    save("augment_pil", **rec)


def gen_sampler():
    """Batch order of the reference's ``InTurnTrainBatchSampler`` / ``InTurnTestBatchSampler`` (data_loader/inTurnLoader.py:15-79) for
    seeded runs.  The module itself does not import here (its imports pull in torchvision); the two classes are pure Python over
    ``random``, so their definitions are compiled from the reference file's AST at generation time -- nothing of the source is
    stored, only the batches they yield."""
    import ast
    from typing import List                      # noqa: F401  (names the class bodies use)
    from torch.utils.data import Sampler         # noqa: F401
    src = open(os.path.join(REF, "data_loader", "inTurnLoader.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("InTurnTrainBatchSampler", "InTurnTestBatchSampler")]
    ns = {"List": List, "Sampler": Sampler, "random": random}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "inTurnLoader.py", "exec"), ns)
    sizes = [23, 31, 19, 40]
    base = [list(range(100 * m, 100 * m + n)) for m, n in enumerate(sizes)]
    rec = dict(sizes=np.array(sizes), batch_size=4)
    for shuffle in (0, 1):
        random.seed(2020 + shuffle)
        smp = ns["InTurnTrainBatchSampler"]([list(b) for b in base], 4, bool(shuffle))
        epochs = [list(smp) for _ in range(3)]                                   # three epochs: wraps and reshuffles inside
        rec[f"train_shuffle{shuffle}_n"] = np.array([len(e) for e in epochs])
        rec[f"train_shuffle{shuffle}"] = np.array([b for e in epochs for b in e], dtype=np.int64)
        rec[f"train_shuffle{shuffle}_len"] = len(smp)
    tst = ns["InTurnTestBatchSampler"]([list(b) for b in base], 4)
    rec["test_flat"] = np.array([i for b in tst for i in b], dtype=np.int64)
    rec["test_sizes"] = np.array([len(b) for b in tst], dtype=np.int64)
    rec["test_len"] = len(tst)
    save("sampler", **rec)


def gen_unet_batch():
    """The reference's DEFAULT constructor arguments: ``UNet(in_ch, out_ch, base_width)`` = norm_type='batch', act_type='relu'
    (network/unet.py:14-15; its own smoke block :35-41 runs them).  Train-mode forward (batch statistics), DiceCE, backward; the
    running statistics two BatchNorm layers hold afterwards; then an eval-mode forward on the updated running statistics."""
    seed, B, H, ncls, w = 29, 3, 32, 3, 8
    net = UNet(1, ncls, w)
    sd = recipe.fill(recipe.unet_shapes(1, ncls, w), seed)
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(("running_" in k or "num_batches" in k) for k in missing.missing_keys)
    net.train()
    x = recipe.synth_images((B, 1, H, H), seed + 1)
    y = recipe.synth_labels(B, H, H, ncls, seed + 2, block=8)
    out = net(x)
    loss = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)(out, y)
    loss.backward()
    st = net.state_dict()
    net.eval()
    with torch.no_grad():
        out_eval = net(x)
    save("unet_batch", seed=seed, B=B, H=H, ncls=ncls, w=w, logits=npy(out), loss=loss.item(), logits_eval=npy(out_eval),
         rm_pre=npy(st["encoder.pre_bn.running_mean"]), rv_pre=npy(st["encoder.pre_bn.running_var"]),
         rm_l3=npy(st["encoder.layer3.bn2.running_mean"]), rv_l3=npy(st["encoder.layer3.bn2.running_var"]),
         nbt=int(st["encoder.pre_bn.num_batches_tracked"]),
         **grad_summary(net, ("encoder.pre_conv.weight", "decoder.fc.weight", "encoder.layer1.bn1.weight", "decoder.layer4.conv1.weight")))


if __name__ == "__main__":
    random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)
    which = sys.argv[1:] or ["unet_small", "unet_relu", "unet_256", "disc_small", "ugan_small", "losses", "iter_small", "networks_zoo", "siblings"]
    for w in which:
        globals()["gen_" + w]()
