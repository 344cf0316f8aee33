"""``PatchNCELoss`` (reference network/patchnce.py:6-51) as one fused HIP kernel pair."""
import torch.nn as nn

from .. import ops

NCE_T = 0.07      # patchnce.py:46


class PatchNCELoss(nn.Module):
    def __init__(self, batch_size):
        super().__init__()
        self.batch_size = batch_size

    def forward(self, feat_q, feat_k):
        """Per-row loss [rows].  Negatives are drawn from groups of ``rows / batch_size`` consecutive rows
        (patchnce.py:32-38) -- with the reference trainer's batch_size=8 fed 16 images, a group spans two
        images; that quirk is reproduced, not fixed."""
        rows = feat_q.shape[0]
        if rows % self.batch_size:
            raise RuntimeError(f"PatchNCELoss: {rows} rows not divisible by batch_size={self.batch_size}")
        return ops.patch_nce(feat_q, feat_k, rows // self.batch_size, NCE_T)
