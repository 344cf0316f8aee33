"""``DiceAndCrossEntropyLoss`` / ``SoftDiceLoss`` (reference misc/loss.py:8-63) on one fused HIP pass."""
import torch
import torch.nn as nn

from .. import ops


class DiceAndCrossEntropyLoss(nn.Module):
    def __init__(self, weight_ce=1.0, weight_dc=1.0, batch_dice=False, process_group=None):
        super().__init__()
        self.weight_ce, self.weight_dc, self.batch_dice = weight_ce, weight_dc, batch_dice
        self.process_group = process_group      # MI355X addition: global-batch Dice under data parallelism

    def forward(self, x, y):
        return ops.dice_ce(x, y, self.weight_ce, self.weight_dc, self.batch_dice, self.process_group)

    # ---- the two stages separately, for trainers that split a captured step at the statistics all-reduce
    # (batch_dice sums tp / fp / fn over the GLOBAL batch, misc/loss.py:52)
    def world(self):
        if self.process_group is None or not self.batch_dice:
            return 1
        import torch.distributed as dist
        return dist.get_world_size(self.process_group)

    def stats(self, x, y):
        """Stage 1 -> ONE flat fp32 tensor [G*C*3 + 1] = {tp, sum_p, count} per class, then the CE sum (local values)."""
        st, ce = ops.dice_ce_stats(x, y, self.batch_dice)
        return torch.cat([st.reshape(-1), ce])

    def reduce_stats(self, flats):
        """Sum the flat statistics of any number of loss terms over the process group with ONE all-reduce (in place)."""
        from .. import parallel
        if self.world() > 1 or (parallel.force_dist() and self.process_group is not None and self.batch_dice):
            ops.all_reduce_dice_stats([tuple(flats)], self.process_group)

    def from_stats(self, x, y, flat):
        """Stage 2: the loss from (global) statistics; differentiable w.r.t. ``x``."""
        g = 1 if self.batch_dice else x.shape[0]
        st, ce = flat[:-1].view(g, x.shape[1], 3), flat[-1:]
        return ops.dice_ce_from_stats(x, y, st, ce, self.weight_ce, self.weight_dc, self.world())


class SoftDiceLoss(nn.Module):
    def __init__(self, batch_dice=False, smooth=1e-5):
        super().__init__()
        if smooth != 1e-5:
            raise NotImplementedError("smooth is fixed at the reference's 1e-5 (misc/loss.py:40)")
        self.batch_dice = batch_dice

    def forward(self, x, y):
        return ops.dice_ce(x, y, 0.0, 1.0, self.batch_dice, None)
