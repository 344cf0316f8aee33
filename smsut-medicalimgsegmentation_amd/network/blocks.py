"""HIP-backed counterparts of the reference's ``network/blocks.py`` building blocks.

Same constructor signatures, attribute names and ``state_dict`` keys/shapes; ``forward`` runs the
gfx950 kernels through ``ops`` (no torch.nn compute).  Only ``norm_type='instance'`` is on the hot
path (every trainer passes it: unetTrainer.py:42, ugan.py:89-96); ``'batch'`` raises.
"""
import math

import torch
import torch.nn as nn

from .. import ops

LRELU_SLOPE = 1e-2


class Conv2d(nn.Module):
    """Parameter holder for nn.Conv2d: ``weight`` is logical OIHW over [KH][KW][I][O] memory."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        self.in_channels, self.out_channels = in_ch, out_ch
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(ops.new_weight(out_ch, in_ch, kernel_size, kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_ch))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # nn.Conv2d defaults: kaiming_uniform(a=sqrt(5)) weight, U(+-1/sqrt(fan_in)) bias
        fan_in = self.in_channels * self.kernel_size * self.kernel_size
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, stats=False):
        """``stats=True`` when the output goes straight into an InstanceNorm2d (fused statistics epilogue)."""
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, stats)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}"


class ConvTranspose2x2(nn.Module):
    """nn.ConvTranspose2d(in, out, kernel_size=2, stride=2, bias=False); weight logical [in, out, 2, 2]."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.in_channels, self.out_channels = in_ch, out_ch
        self.weight = nn.Parameter(ops.new_convT_weight(in_ch, out_ch))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x):
        return ops.conv_transpose2x2(x, self.weight)


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(C, affine=True): keys ``weight`` / ``bias`` only (no running stats)."""

    def __init__(self, channels):
        super().__init__()
        self.num_features = channels
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x, slope=None):
        """``slope`` fuses the following LeakyReLU/ReLU into the same kernel."""
        return ops.instnorm_act(x, self.weight, self.bias, slope)


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d(C) -- the reference's DEFAULT norm (``norm_type='batch'``: network/unet.py:14-15, blocks.py:19-26;
    no trainer uses it).  Same state_dict keys (``weight``, ``bias``, ``running_mean``, ``running_var``,
    ``num_batches_tracked``), eps 1e-5, momentum 0.1.

    Train mode: batch statistics over (N, H, W) per channel ARE instance statistics of the batch viewed as ONE instance of
    N*H*W pixels -- NHWC memory makes that a free view -- so forward, backward and double backward are the InstanceNorm kernels
    (``ops.InstNormActFn``) with N' = 1; the running statistics are updated from the kernel's own mean / rstd (unbiased variance,
    as torch).  Eval mode: a per-channel affine on the running statistics, run as a diagonal 1x1 conv + bias through the HIP
    conv path (cold path: kept simple)."""

    def __init__(self, channels, eps=1e-5, momentum=0.1):
        super().__init__()
        if abs(eps - ops.IN_EPS) > 1e-12:
            raise NotImplementedError("BatchNorm2d: eps is fixed to 1e-5 (the kernels' InstanceNorm epsilon)")
        self.num_features, self.eps, self.momentum = channels, eps, momentum
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, slope=None):
        n, c, h, w = x.shape
        if self.training:
            xc = ops.cl(x)
            one = xc.permute(0, 2, 3, 1).reshape(1, n * h, w, c).permute(0, 3, 1, 2)       # [1, C, N*H, W], same NHWC memory
            y, mean, rstd = ops.InstNormActFn.apply(one, self.weight, self.bias, 0.0 if slope is None else slope, slope is not None)
            with torch.no_grad():
                cnt = n * h * w
                var = (1.0 / (rstd[0] * rstd[0]) - self.eps).clamp_min_(0.0)                # biased batch variance
                self.running_mean.lerp_(mean[0], self.momentum)
                self.running_var.lerp_(var * (cnt / max(cnt - 1, 1)), self.momentum)
                self.num_batches_tracked += 1
            return y.permute(0, 2, 3, 1).reshape(n, h, w, c).permute(0, 3, 1, 2)
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        shift = self.bias - self.running_mean * scale
        wd = ops.new_weight(c, c, 1, 1, device=x.device)
        with torch.no_grad():
            wd.zero_()
        wd = wd + torch.diag(scale).view(c, c, 1, 1)                                      # (keeps the graph to weight / bias)
        y = ops.conv2d(x, wd, shift, stride=1, pad=0)
        return y if slope is None else ops.leaky_relu(y, slope)


class Act(nn.Module):
    """LeakyReLU(slope) / ReLU (slope 0) marker; blocks read ``.slope`` and fuse it into the norm kernel."""

    def __init__(self, slope):
        super().__init__()
        self.slope = float(slope)

    def forward(self, x):
        return ops.leaky_relu(x, self.slope)


class Upsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) (parameter-free)."""

    def forward(self, x):
        return ops.bilinear_up2(x)


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    if groups != 1 or dilation != 1:
        raise NotImplementedError("hot path uses groups=1, dilation=1 only (reference blocks.py:10-12)")
    return Conv2d(in_planes, out_planes, 3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, 1, stride=stride, padding=0, bias=False)


def get_norm(channels, norm_type):
    if norm_type == "instance":
        return InstanceNorm2d(channels)
    if norm_type == "batch":
        return BatchNorm2d(channels)
    raise NotImplementedError


def get_act(act_type, inplace=True, negative=1e-2):
    if act_type == "relu":
        return Act(0.0)
    if act_type == "lrelu":
        return Act(negative)
    raise NotImplementedError


class UpSampleAndConcat(nn.Module):
    def __init__(self, in_ch, out_ch, transposed=True):
        super().__init__()
        if transposed:
            self.up = ConvTranspose2x2(in_ch, out_ch)
        else:
            self.up = nn.Sequential(Upsample2x(), conv1x1(in_ch, out_ch))      # keys: up.1.weight

    def forward(self, x, skip):
        if isinstance(self.up, nn.Sequential):
            # bilinear interpolation (per channel, spatial) and a 1x1 conv (per pixel, across channels) are both
            # linear and commute exactly in real arithmetic: run the conv at the LOW resolution (4x fewer FLOPs
            # and bytes), then upsample its half-as-wide output.  fp32 rounding differs at the 1e-7 level.
            return ops.concat_channels_deferred(self.up[0](self.up[1](x)), skip)
        return ops.concat_channels_deferred(self.up(x), skip)


class BasicBlock(nn.Module):
    def __init__(self, in_ch, out_ch, norm, act, **kwargs):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.bn1 = get_norm(out_ch, norm)
        self.relu = get_act(act)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.bn2 = get_norm(out_ch, norm)
        self.downsample = in_ch != out_ch
        if self.downsample:
            self.shortcut1 = conv1x1(in_ch, out_ch)
            self.shortcut2 = get_norm(out_ch, norm)

    def forward(self, x):
        s = self.relu.slope
        ws = self.shortcut1.weight if self.downsample else None
        fused_ok = isinstance(self.bn1, InstanceNorm2d)        # (the fused block kernels ARE InstanceNorm; BatchNorm: op by op)
        if isinstance(x, ops.CatParts):            # cat([up, skip]) not built yet (UpSampleAndConcat)
            if fused_ok and ops.basic_block_cat_fusable(x, self.conv1.weight, ws):
                return ops.basic_block_cat(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                           self.bn2.weight, self.bn2.bias, ws, self.shortcut2.weight, self.shortcut2.bias, s)
            x = x.tensor()
        if fused_ok and ops.basic_block_fusable(x, self.conv1.weight, ws):
            sc = (ws, self.shortcut2.weight, self.shortcut2.bias) if self.downsample else (None, None, None)
            return ops.basic_block(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                   self.bn2.weight, self.bn2.bias, *sc, s)
        # op-by-op path (channel counts the MFMA kernels do not cover; also the cross-check for the fused one)
        y = self.bn1(self.conv1(x, stats=True), slope=s)
        y = self.bn2(self.conv2(y, stats=True))
        idn = self.shortcut2(self.shortcut1(x, stats=True)) if self.downsample else x
        return ops.add_act(y, idn, s)

    def forward_pool(self, x):
        """(max_pool2(out), out) of an encoder level (blocks.py:131-133, ugan.py:36-39) with the block's residual tail and the pooling
        as ONE pass (r05, ``ops.basic_block_pool``) -- or None where that form does not apply (the caller then runs block + pooling)."""
        if not (self.downsample and isinstance(self.bn1, InstanceNorm2d)) or isinstance(x, ops.CatParts):
            return None
        ws = self.shortcut1.weight
        if not ops.basic_block_pool_fusable(x, self.conv1.weight, ws):
            return None
        out, pooled = ops.basic_block_pool(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight,
                                           self.bn2.bias, ws, self.shortcut2.weight, self.shortcut2.bias, self.relu.slope)
        return pooled, out


class BottleBlock(nn.Module):
    def __init__(self, in_channels, out_channels, norm_type="batch", act_type="relu", stride=1):
        super().__init__()
        assert stride in (1, 2)
        self.conv1 = conv3x3(in_channels, out_channels)
        self.bn1 = get_norm(out_channels, norm_type)
        self.relu = get_act(act_type)
        self.conv2 = conv3x3(out_channels, out_channels)
        self.bn2 = get_norm(out_channels, norm_type)
        self.stride = stride
        self.feeds_stride2 = False           # set by the owner: the output goes into ANOTHER stride-2 BottleBlock (conv1 + pooled shortcut)
        self.downsample = None
        if in_channels != out_channels:
            self.downsample = nn.Sequential(conv1x1(in_channels, out_channels), get_norm(out_channels, norm_type))

    def forward(self, x):
        s = self.relu.slope
        idn = x
        if self.stride == 2:
            idn = getattr(x, "_smsut_avg_pooled", None)        # written by the previous block's residual tail (r05, first-order passes)
            if idn is None:
                idn = ops.avg_pool2(x)
        y1 = self.conv1(x, stats=True)
        if self.stride == 2 and isinstance(self.bn1, InstanceNorm2d) and ops.instnorm_act_pool_fusable(y1):
            y = ops.instnorm_act_pool(y1, self.bn1.weight, self.bn1.bias, s)      # first-order passes: IN + act + pool in one pass (r05)
        else:
            y = self.bn1(y1, slope=s)
            if self.stride == 2:
                y = ops.avg_pool2(y)
        y2 = self.conv2(y, stats=True)
        if self.downsample is not None:
            sc = self.downsample[0](idn, stats=True)
            if isinstance(self.bn2, InstanceNorm2d) and ops.res_tail_fusable(y2, sc):          # first-order passes: IN2 + IN(shortcut) + add + act in one kernel
                if self.feeds_stride2 and ops.res_tail_pool_fusable(y2):
                    out, pooled = ops.res_tail_pool(y2, self.bn2.weight, self.bn2.bias, sc, self.downsample[1].weight,
                                                    self.downsample[1].bias, s)
                    out._smsut_avg_pooled = pooled                # the next block's shortcut input, from the same pass
                    return out
                return ops.res_tail(y2, self.bn2.weight, self.bn2.bias, sc, self.downsample[1].weight,
                                    self.downsample[1].bias, s)
            idn = self.downsample[1](sc)
        return ops.add_act(self.bn2(y2), idn, s)


class Encoder(nn.Module):
    def __init__(self, in_ch, block, width=32, norm="batch", act="lrelu", **kwargs):
        super().__init__()
        self.pre_conv = Conv2d(in_ch, width // 2, 5, stride=1, padding=2, bias=False)
        self.pre_bn = get_norm(width // 2, norm)
        self.pre_relu = get_act(act)
        chans = [width // 2, width, 2 * width, 4 * width, 8 * width, 16 * width]
        for i in range(1, 6):
            setattr(self, f"layer{i}", block(chans[i - 1], chans[i], norm, act, **kwargs))
            if i < 5:
                setattr(self, f"pool{i}", MaxPool2x2())

    def forward(self, x):
        skips = []
        x = self.pre_bn(self.pre_conv(x, stats=True), slope=self.pre_relu.slope)
        for i in range(1, 5):
            x, skip = encoder_level(getattr(self, f"layer{i}"), getattr(self, f"pool{i}"), x)
            skips.append(skip)
        return self.layer5(x), skips


def encoder_level(block, pool, x):
    """(pooled, skip) of one encoder level: block -> MaxPool2d(2, 2), the block output also being the skip connection."""
    fused = block.forward_pool(x) if isinstance(block, BasicBlock) else None
    if fused is not None:
        return fused
    return pool.pool_skip(block(x))


class MaxPool2x2(nn.Module):
    """nn.MaxPool2d(2, 2)."""

    def forward(self, x):
        return ops.max_pool2(x)

    def pool_skip(self, x):
        """(pooled, skip): the level's two uses of x in one autograd node (skip gradient summed in the pooling backward)."""
        return ops.max_pool2_skip(x)


class Decoder(nn.Module):
    def __init__(self, out_ch, block, width=32, norm="batch", act="lrelu", **kwargs):
        super().__init__()
        for lvl, m in zip((4, 3, 2, 1), (8, 4, 2, 1)):
            setattr(self, f"up{lvl}", UpSampleAndConcat(2 * m * width, m * width))
            setattr(self, f"layer{lvl}", block(2 * m * width, m * width, norm, act, **kwargs))
        self.fc = conv1x1(width, out_ch)

    def forward(self, x, skips):
        for lvl in (4, 3, 2, 1):
            x = getattr(self, f"layer{lvl}")(getattr(self, f"up{lvl}")(x, skips[lvl - 1]))
        return self.fc(x)


def init_conv_kaiming(module, nonlinearity):
    """The reference's post-construction init loop (unet.py:21-27, ugan.py:100-106,145-151,217-223)."""
    for m in module.modules():
        if isinstance(m, (Conv2d, ConvTranspose2x2)):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity=nonlinearity)
        elif isinstance(m, InstanceNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
