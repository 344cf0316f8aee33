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
