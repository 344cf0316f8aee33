"""ugan unified translation + segmentation networks with the reference's constructor signatures
(network/ugan.py:22-339): ``Encoder``, ``Decoder``, ``UGAN``, ``UGANnce``, ``Discriminator``,
``define_F``, ``PatchSampleF`` -- forward passes run on the gfx950 kernels."""
import numpy as np
import torch
import torch.nn as nn

from .. import config as cfg
from .. import ops
from . import networks
from .blocks import (Act, BasicBlock, BottleBlock, Conv2d, MaxPool2x2, UpSampleAndConcat, encoder_level, get_act, get_norm,
                     init_conv_kaiming)


class _Pre(nn.Sequential):
    """nn.Sequential(conv5x5, norm, act) whose forward fuses the activation into the norm kernel."""

    def forward(self, x):
        return self[1](self[0](x, stats=True), slope=self[2].slope)


class Encoder(nn.Module):
    def __init__(self, in_ch, base_width=32, norm_type="batch", act_type="relu"):
        super().__init__()
        w = base_width
        self.pre = _Pre(Conv2d(in_ch, w // 2, 5, stride=1, padding=2, bias=False), get_norm(w // 2, norm_type),
                        get_act(act_type))
        chans = [w // 2, w, 2 * w, 4 * w, 8 * w]
        for i in range(1, 5):
            setattr(self, f"enc{i}", BasicBlock(chans[i - 1], chans[i], norm_type, act_type))
            setattr(self, f"pool{i}", MaxPool2x2())

    def forward(self, x):
        skips = []
        x = self.pre(x)
        for i in range(1, 5):
            x, skip = encoder_level(getattr(self, f"enc{i}"), getattr(self, f"pool{i}"), x)
            skips.append(skip)
        skips.reverse()                    # deepest first (ugan.py:54)
        return x, skips


class Decoder(nn.Module):
    def __init__(self, out_ch, base_width=32, norm_type="batch", act_type="relu", tranposed=True, use_tanh=False):
        super().__init__()
        w = base_width
        for lvl, m in zip((4, 3, 2, 1), (8, 4, 2, 1)):
            setattr(self, f"up{lvl}", UpSampleAndConcat(2 * m * w, m * w, transposed=tranposed))
            setattr(self, f"dec{lvl}", BasicBlock(2 * m * w, m * w, norm_type, act_type))
        self.fc = Conv2d(w, out_ch, 1, bias=True)
        self.tanh = _Tanh() if use_tanh else None

    def forward(self, e5, x_ens):
        h = e5
        for i, lvl in enumerate((4, 3, 2, 1)):
            h = getattr(self, f"dec{lvl}")(getattr(self, f"up{lvl}")(h, x_ens[i]))
        out = self.fc(h)
        return self.tanh(out) if self.tanh is not None else out


class _Tanh(nn.Module):
    def forward(self, x):
        return ops.tanh(x)


class _UGANBase(nn.Module):
    def _build(self, in_ch, out_ch, n_modal, base_width, with_nce):
        self.n_modal = n_modal
        self.tsl_encoder = Encoder(in_ch + n_modal, base_width, norm_type="instance", act_type="lrelu")
        self.seg_encoder = Encoder(in_ch, base_width, norm_type="instance", act_type="lrelu")
        self.enc5 = BasicBlock(8 * base_width, 16 * base_width, norm="instance", act="lrelu")
        if with_nce:
            self.netF = define_F(in_ch)
            if not self.netF.mlp_init:
                self.netF.create_mlp(cfg.nce_layers)
        self.tsl_decoder = Decoder(1, base_width, norm_type="instance", act_type="lrelu", tranposed=False,
                                   use_tanh=True)
        self.seg_decoder = Decoder(out_ch, base_width, norm_type="instance", act_type="lrelu", tranposed=True,
                                   use_tanh=False)
        # ugan.py:100-106 / :145-151 -- note this re-initialises netF's Linear?  No: the reference loop only
        # matches nn.Conv2d / nn.ConvTranspose2d / norms, so netF keeps init_weights('normal', 0.02).
        init_conv_kaiming(self, "leaky_relu")

    # The generator is two independent branches that share only the ``enc5`` module (ugan.py:156-170): the TRANSLATION branch
    # (x + modality planes -> tsl_encoder -> enc5 -> tsl_decoder) and the SEGMENTATION branch (x -> seg_encoder -> enc5 ->
    # seg_decoder).  ``_trunk`` is the reference's forward; the trainers may run the branches separately (the segmentation branch
    # of the two generator passes of an iteration as ONE batched pass: InstanceNorm is per sample, so the values are the same).
    def _tsl_branch(self, x, m):
        if m is None:
            m = torch.zeros(x.size(0), self.n_modal, device=x.device)
        tsl_in = ops.modal_planes(x, m.to(x.device))            # ugan.py:156-159
        t_bot, t_sk = self.tsl_encoder(tsl_in)
        t_e5 = self.enc5(t_bot)                                 # shared enc5 (ugan.py:163,168)
        return self.tsl_decoder(t_e5, t_sk), t_e5

    def _seg_branch(self, x):
        s_bot, s_sk = self.seg_encoder(x)
        return self.seg_decoder(self.enc5(s_bot), s_sk)

    def _trunk(self, x, m):
        tsl, t_e5 = self._tsl_branch(x, m)
        return self._seg_branch(x), tsl, t_e5


class UGAN(_UGANBase):
    def __init__(self, in_ch, out_ch, n_modal, base_width=32):
        super().__init__()
        self._build(in_ch, out_ch, n_modal, base_width, with_nce=False)

    def forward(self, x, m=None):
        seg, tsl, _ = self._trunk(x, m)
        return seg, tsl


class UGANnce(_UGANBase):
    def __init__(self, in_ch, out_ch, n_modal, base_width=32, val_phase=False):
        super().__init__()
        self.val_phase = val_phase
        self._build(in_ch, out_ch, n_modal, base_width, with_nce=True)

    def forward(self, x, m=None, sample_ids=None, val_phase=False, branch=None):
        """Reference contract (ugan.py:153-195) for ``branch=None``.  ``branch`` (this package's trainers only): "tsl" -> the
        translation branch alone, ``(tsl, feat_pool, sample_ids, t_e5)``; "seg" -> the segmentation branch alone, ``seg``."""
        if branch == "seg":
            return self._seg_branch(x)
        if branch == "tsl":
            tsl, t_e5 = self._tsl_branch(x, m)
            if sample_ids is None:
                return tsl, None, None, t_e5
            feat_pool, _ = self.netF([t_e5], patch_ids=sample_ids)
            return tsl, feat_pool, sample_ids, t_e5
        seg, tsl, t_e5 = self._trunk(x, m)
        if val_phase:
            return seg, tsl
        feats = [t_e5]
        if sample_ids is None:
            feat_pool, sample_ids = self.netF(feats)
        else:
            feat_pool, _ = self.netF(feats, patch_ids=sample_ids)
        return seg, tsl, feat_pool, sample_ids


class Discriminator(nn.Module):
    def __init__(self, input_size, n_modal, base_width=32, max_width=512):
        super().__init__()
        layers = [Conv2d(1, base_width, 4, stride=2, padding=1, bias=True), Act(0.01)]   # nn.LeakyReLU() default slope
        repeat_num = int(np.log2(input_size)) - 2
        in_w = base_width
        out_w = base_width
        for _ in range(1, repeat_num):
            out_w = min(in_w * 2, max_width)
            layers.append(BottleBlock(in_w, out_w, norm_type="instance", act_type="lrelu", stride=2))
            in_w = out_w
        blocks_ = [m for m in layers if isinstance(m, BottleBlock)]
        for a, b in zip(blocks_[:-1], blocks_[1:]):          # consecutive stride-2 blocks: the first one's tail also writes the pooled
            a.feeds_stride2 = b.stride == 2                  # shortcut input of the second (ops.res_tail_pool)
        self.main = nn.Sequential(*layers)
        k = int(input_size / np.power(2, repeat_num))
        self.conv_src = Conv2d(out_w, 1, 3, stride=1, padding=1, bias=False)
        self.conv_cls = Conv2d(out_w, n_modal, k, bias=False)
        init_conv_kaiming(self, "leaky_relu")

    def forward(self, x):
        h = self.main(x)
        out_src = self.conv_src(h)
        out_cls = self.conv_cls(h)
        return out_src, out_cls.reshape(out_cls.size(0), out_cls.size(1))


def define_F(input_nc, netF="mlp_sample", norm="batch", use_dropout=False, init_type="normal", init_gain=0.02,
             no_antialias=False, gpu_ids=None, netF_nc=256):
    net = PatchSampleF(use_mlp=True, init_type=init_type, init_gain=init_gain, gpu_ids=gpu_ids or [], nc=netF_nc)
    return init_net(net, init_type, init_gain, gpu_ids or [])


def init_net(net, init_type="normal", init_gain=0.02, gpu_ids=(), debug=False, initialize_weights=True):
    return networks.init_net(net, init_type, init_gain, gpu_ids, debug, initialize_weights)


class PatchSampleF(nn.Module):
    def __init__(self, use_mlp=False, init_type="normal", init_gain=0.02, nc=256, gpu_ids=()):
        super().__init__()
        self.l2norm = networks.Normalize(2)
        self.use_mlp = use_mlp
        self.nc = nc
        self.mlp_init = False
        self.init_type, self.init_gain, self.gpu_ids = init_type, init_gain, list(gpu_ids)

    def create_mlp(self, nce_layers, input_nc=256):
        for mlp_id, _ in enumerate(nce_layers):
            mlp = nn.Sequential(networks.Linear(input_nc, self.nc), networks.ReLU(), networks.Linear(self.nc, self.nc))
            setattr(self, "mlp_%d" % mlp_id, mlp)
        init_net(self, self.init_type, self.init_gain, self.gpu_ids)
        self.mlp_init = True

    def forward(self, feats, num_patches=64, patch_ids=None):
        """ugan.py:302-339: the SAME patch ids for every image of the batch; ids drawn with torch.randperm
        on the feature's device when not given."""
        return_ids, return_feats = [], []
        if self.use_mlp and not self.mlp_init:
            self.create_mlp(cfg.nce_layers)
        for feat_id, feat in enumerate(feats):
            hw = feat.shape[2] * feat.shape[3]
            if num_patches <= 0:
                raise NotImplementedError("num_patches=0 (dense) is not used by any trainer")
            if patch_ids is not None:
                patch_id = patch_ids[feat_id]
            else:
                patch_id = torch.randperm(hw, device=feat.device)[: int(min(num_patches, hw))]
            x_sample = ops.gather_patches(feat, patch_id.to(feat.device))
            if self.use_mlp:
                x_sample = getattr(self, "mlp_%d" % feat_id)(x_sample)
            return_ids.append(patch_id)
            return_feats.append(self.l2norm(x_sample))
        return return_feats, return_ids
