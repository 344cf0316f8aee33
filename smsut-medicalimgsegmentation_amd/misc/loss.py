"""``DiceAndCrossEntropyLoss`` / ``SoftDiceLoss`` (reference misc/loss.py:8-63) on one fused HIP pass."""
import torch.nn as nn

from .. import ops


class DiceAndCrossEntropyLoss(nn.Module):
    def __init__(self, weight_ce=1.0, weight_dc=1.0, batch_dice=False, process_group=None):
        super().__init__()
        self.weight_ce, self.weight_dc, self.batch_dice = weight_ce, weight_dc, batch_dice
        self.process_group = process_group      # MI355X addition: global-batch Dice under data parallelism

    def forward(self, x, y):
        return ops.dice_ce(x, y, self.weight_ce, self.weight_dc, self.batch_dice, self.process_group)


class SoftDiceLoss(nn.Module):
    def __init__(self, batch_dice=False, smooth=1e-5):
        super().__init__()
        if smooth != 1e-5:
            raise NotImplementedError("smooth is fixed at the reference's 1e-5 (misc/loss.py:40)")
        self.batch_dice = batch_dice

    def forward(self, x, y):
        return ops.dice_ce(x, y, 0.0, 1.0, self.batch_dice, None)
