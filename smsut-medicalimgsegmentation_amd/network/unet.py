"""``UNet`` with the reference's constructor signature (network/unet.py:13-32) on gfx950 kernels.

The class is a thin composition: the attribute names ``encoder`` / ``decoder`` (and everything below them in
``blocks.py``) reproduce the reference's ``state_dict`` keys and tensor shapes, so its checkpoints load unchanged;
the modules themselves hold HWIO-memory weights and launch the HIP kernels of ``ops`` (fused BasicBlock: two 3x3 MFMA
convolutions with statistics epilogues, one normalise+activation pass, one residual-tail pass).  Inputs are NCHW
tensors on a HIP device; outputs are NCHW with channels_last memory.  BatchNorm (``norm_type='batch'``) is not
implemented -- no trainer of the reference uses it -- and raises in ``blocks.make_norm``.
"""
import torch.nn as nn

from .blocks import BasicBlock, Decoder, Encoder, init_conv_kaiming


class UNet(nn.Module):
    def __init__(self, in_ch, out_ch, base_width=64, norm_type="batch", act_type="relu"):
        super().__init__()
        self.encoder = Encoder(in_ch, BasicBlock, base_width, norm=norm_type, act=act_type)
        self.decoder = Decoder(out_ch, BasicBlock, base_width, norm=norm_type, act=act_type)
        init_conv_kaiming(self, "relu" if act_type == "relu" else "leaky_relu")

    def forward(self, x):
        x, skips = self.encoder(x)
        return self.decoder(x, skips)
