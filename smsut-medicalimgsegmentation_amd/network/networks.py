"""The symbols of the reference's ``network/networks.py`` that any trainer reaches:
``Normalize`` (:234-243) and ``init_weights`` (:163-195).  The vendored CUT/MUNIT zoo around them is
dead code upstream; ``ResnetGenerator`` / ``NLayerDiscriminator`` are the "next" rows.
"""
import torch.nn as nn
from torch.nn import init

from .. import ops
from . import blocks


class Normalize(nn.Module):
    def __init__(self, power=2):
        super().__init__()
        if power != 2:
            raise NotImplementedError("only the L2 form (power=2) is used on the hot path (ugan.py:274)")
        self.power = power

    def forward(self, x):
        return ops.l2_normalize(x)


class Linear(nn.Module):
    """nn.Linear parameter holder: ``weight`` logical [out, in] over [in][out] memory (a 1x1 conv in HWIO)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        import math
        import torch
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(ops.new_linear_weight(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features)
        init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class ReLU(nn.Module):
    def forward(self, x):
        p, c = x.shape
        return ops.leaky_relu(x.reshape(p, c, 1, 1), 0.0).reshape(p, c)


def init_weights(net, init_type="normal", init_gain=0.02, debug=False):
    """networks.py:163-195: Conv*/Linear weights by ``init_type``, biases 0.  (The BatchNorm2d branch of
    the reference never fires on the hot path -- there is no BatchNorm in any reachable module.)"""

    def init_func(m):
        if isinstance(m, (blocks.Conv2d, blocks.ConvTranspose2x2, Linear)):
            if init_type == "normal":
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if getattr(m, "bias", None) is not None:
                init.constant_(m.bias.data, 0.0)

    net.apply(init_func)


def init_net(net, init_type="normal", init_gain=0.02, gpu_ids=(), debug=False, initialize_weights=True):
    """networks.py:198-214."""
    if len(gpu_ids) > 0:
        net.to(gpu_ids[0])
    if initialize_weights:
        init_weights(net, init_type, init_gain=init_gain, debug=debug)
    return net


# =====================================================================================================================
# the CUT-style generator / PatchGAN discriminator that BASELINE.json's north_star names
# (networks.py:607-702, 815-872, 977-1032, 1067-1080).  Dead code upstream (no trainer instantiates them); built here as
# compositions of the same HIP kernels so the constructor signatures and state_dict keys exist and are parity-tested.
# The 7x7 and 4x4 convs run on the shape-complete direct kernels this round (MFMA tiles cover 1x1 / 3x3 only).
import functools  # noqa: E402

import torch  # noqa: E402


class InstanceNorm2dNA(nn.Module):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) (get_norm_layer('instance'), networks.py:124-125):
    no parameters, no buffers in the state_dict."""

    def __init__(self, channels, **_):
        super().__init__()
        self.num_features = channels
        self.register_buffer("_one", torch.ones(channels), persistent=False)
        self.register_buffer("_zero", torch.zeros(channels), persistent=False)

    def forward(self, x, slope=None):
        return ops.instnorm_act(x, self._one, self._zero, slope)


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type="instance"):
    """networks.py:113-131 ('batch' -> blocks.BatchNorm2d: affine, running statistics, as nn.BatchNorm2d)."""
    if norm_type == "instance":
        return functools.partial(InstanceNorm2dNA)
    if norm_type == "none":
        return lambda ch: Identity()
    if norm_type == "batch":
        return functools.partial(blocks.BatchNorm2d)
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def _is_instance_norm(norm_layer):
    f = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    return f in (InstanceNorm2dNA, nn.InstanceNorm2d)


def _make_norm(norm_layer, ch):
    f = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    if f in (InstanceNorm2dNA, nn.InstanceNorm2d):
        kw = norm_layer.keywords if isinstance(norm_layer, functools.partial) else {}
        if kw.get("affine", False):
            return blocks.InstanceNorm2d(ch)
        return InstanceNorm2dNA(ch)
    if f in (nn.BatchNorm2d, blocks.BatchNorm2d):        # the reference's default norm_layer (networks.py:613, :980)
        return blocks.BatchNorm2d(ch)
    return norm_layer(ch)


def get_filter(filt_size=3):
    import numpy as np
    a = {1: [1.], 2: [1., 1.], 3: [1., 2., 1.], 4: [1., 3., 3., 1.], 5: [1., 4., 6., 4., 1.]}[filt_size]
    a = np.array(a)
    f = torch.Tensor(a[:, None] * a[None, :])
    return f / torch.sum(f)


class ReflectionPad2d(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        return ops.pad2d(x, (self.p,) * 4, "reflect")


class Downsample(nn.Module):
    """Anti-aliased stride-2 down-sampling (networks.py:37-60); only the defaults (reflect, filt 3, stride 2) are built."""

    def __init__(self, channels, pad_type="reflect", filt_size=3, stride=2, pad_off=0):
        super().__init__()
        if (pad_type not in ("reflect", "refl")) or filt_size != 3 or stride != 2 or pad_off != 0:
            raise NotImplementedError("Downsample: only pad_type='reflect', filt_size=3, stride=2, pad_off=0")
        self.channels = channels
        self.register_buffer("filt", get_filter(3)[None, None].repeat((channels, 1, 1, 1)))      # state_dict key parity

    def forward(self, x):
        return ops.blur_down2(x)


class Upsample(nn.Module):
    """networks.py:73-93 with the defaults (replicate pad, filt 4, stride 2).  The depthwise transposed conv with
    [1,3,3,1]^2/16 after a replicate pad and the two crops is exactly x2 bilinear interpolation with clamped borders
    (out[2m] = (x[m-1] + 3x[m])/4, out[2m+1] = (3x[m] + x[m+1])/4), i.e. the kernel behind nn.Upsample(bilinear)."""

    def __init__(self, channels, pad_type="repl", filt_size=4, stride=2):
        super().__init__()
        if pad_type not in ("repl", "replicate") or filt_size != 4 or stride != 2:
            raise NotImplementedError("Upsample: only pad_type='repl', filt_size=4, stride=2")
        self.channels = channels
        self.register_buffer("filt", (get_filter(4) * 4)[None, None].repeat((channels, 1, 1, 1)))

    def forward(self, x):
        return ops.bilinear_up2(x)


class ResnetBlock(nn.Module):
    """networks.py:815-872: x + [pad, conv3x3, norm, ReLU, pad, conv3x3, norm](x)."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        if use_dropout:
            raise NotImplementedError("use_dropout=True is not on any configured path")
        if padding_type not in ("reflect", "replicate", "zero"):
            raise NotImplementedError("padding [%s] is not implemented" % padding_type)
        self.padding_type = padding_type
        seq = []
        for k in range(2):
            if padding_type != "zero":
                seq.append(_Pad(padding_type, 1))
            seq.append(blocks.Conv2d(dim, dim, 3, padding=1, bias=use_bias))
            seq.append(_make_norm(norm_layer, dim))
            if k == 0:
                seq.append(blocks.Act(0.0))
        self.conv_block = nn.Sequential(*seq)

    def forward(self, x):
        h = x
        mods = list(self.conv_block)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _Pad):
                # "pad then valid conv" == same-padded conv on the padded tensor, cropped by 1: keeps the conv on the
                # MFMA stride-1 'same' kernel; the crop is a window copy
                h = ops.pad2d(mods[i + 1](m(h)), (-1, -1, -1, -1))
                i += 2
            elif isinstance(m, (InstanceNorm2dNA, blocks.InstanceNorm2d)) and i + 1 < len(mods) and isinstance(mods[i + 1], blocks.Act):
                h = m(h, slope=mods[i + 1].slope)
                i += 2
            else:
                h = m(h)
                i += 1
        return ops.add_act(x, h, 1.0)            # slope 1 == identity: plain residual add


class _Pad(nn.Module):
    def __init__(self, mode, p):
        super().__init__()
        self.mode, self.p = mode, p

    def forward(self, x):
        return ops.pad2d(x, (self.p,) * 4, self.mode)


class _Seq(nn.Sequential):
    """nn.Sequential that fuses (norm, activation) pairs into one kernel and accepts the reference's layer taps."""

    def run(self, x, start=0, stop=None, tap=None):
        mods = list(self)
        i = start
        stop = len(mods) if stop is None else stop
        while i < stop:
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < stop else None
            fuse = (isinstance(m, (InstanceNorm2dNA, blocks.InstanceNorm2d)) and isinstance(nxt, blocks.Act)
                    and (tap is None or i not in tap))
            if fuse:
                x = m(x, slope=nxt.slope)
                i += 1
            else:
                x = m(x)
            if tap is not None and i in tap:
                tap[i] = x
            i += 1
        return x


class ResnetGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False, n_blocks=6,
                 padding_type="reflect", no_antialias=False, no_antialias_up=False, opt=None):
        assert n_blocks >= 0
        super().__init__()
        self.opt = opt
        if no_antialias or no_antialias_up:
            raise NotImplementedError("the strided-conv / ConvTranspose3x3 variants (no_antialias*) are not built")
        use_bias = _is_instance_norm(norm_layer)
        model = [ReflectionPad2d(3), blocks.Conv2d(input_nc, ngf, 7, padding=0, bias=use_bias), _make_norm(norm_layer, ngf),
                 blocks.Act(0.0)]
        for i in range(2):
            mult = 2 ** i
            model += [blocks.Conv2d(ngf * mult, ngf * mult * 2, 3, stride=1, padding=1, bias=use_bias),
                      _make_norm(norm_layer, ngf * mult * 2), blocks.Act(0.0), Downsample(ngf * mult * 2)]
        for _ in range(n_blocks):
            model += [ResnetBlock(ngf * 4, padding_type=padding_type, norm_layer=norm_layer, use_dropout=use_dropout,
                                  use_bias=use_bias)]
        for i in range(2):
            mult = 2 ** (2 - i)
            model += [Upsample(ngf * mult), blocks.Conv2d(ngf * mult, int(ngf * mult / 2), 3, stride=1, padding=1, bias=use_bias),
                      _make_norm(norm_layer, int(ngf * mult / 2)), blocks.Act(0.0)]
        model += [ReflectionPad2d(3), blocks.Conv2d(ngf, output_nc, 7, padding=0, bias=True), _TanhM()]
        self.model = _Seq(*model)

    def forward(self, input, layers=[], encode_only=False):
        if -1 in layers:
            layers.append(len(self.model))
        if len(layers) > 0:
            tap = {i: None for i in layers}
            last = layers[-1] if encode_only else len(self.model) - 1
            out = self.model.run(input, 0, min(last + 1, len(self.model)), tap)
            feats = [tap[i] for i in layers if tap.get(i) is not None]
            if encode_only:
                return feats
            return out, feats
        return self.model.run(input)


class _TanhM(nn.Module):
    def forward(self, x):
        return ops.tanh(x)


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, no_antialias=False):
        super().__init__()
        if no_antialias:
            raise NotImplementedError("no_antialias=True (stride-2 4x4 convs) is not built")
        use_bias = _is_instance_norm(norm_layer)
        kw, padw = 4, 1
        seq = [blocks.Conv2d(input_nc, ndf, kw, stride=1, padding=padw, bias=True), blocks.Act(0.2), Downsample(ndf)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [blocks.Conv2d(ndf * nf_prev, ndf * nf_mult, kw, stride=1, padding=padw, bias=use_bias),
                    _make_norm(norm_layer, ndf * nf_mult), blocks.Act(0.2), Downsample(ndf * nf_mult)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [blocks.Conv2d(ndf * nf_prev, ndf * nf_mult, kw, stride=1, padding=padw, bias=use_bias),
                _make_norm(norm_layer, ndf * nf_mult), blocks.Act(0.2)]
        seq += [blocks.Conv2d(ndf * nf_mult, 1, kw, stride=1, padding=padw, bias=True)]
        self.model = _Seq(*seq)

    def forward(self, input):
        return self.model.run(input)


class PatchDiscriminator(NLayerDiscriminator):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, no_antialias=False):
        super().__init__(input_nc, ndf, 2, norm_layer, no_antialias)

    def forward(self, input):
        B, C, H, W = input.shape
        size = 16
        Y, X = H // size, W // size
        input = input.reshape(B, C, Y, size, X, size).permute(0, 2, 4, 1, 3, 5).contiguous().view(B * Y * X, C, size, size)
        return super().forward(input)
