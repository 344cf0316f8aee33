"""``meanTeacherTrainer`` (reference trainer/meanTeacherTrainer.py:37-151): student U-Net + EMA teacher, DiceCE on
the labeled half, softmax-MSE consistency between the student on the unlabeled half and the teacher on a noised
copy (from iteration 100 on), SGD + poly LR, EMA update after every step.  Same kernels as the U-Net path plus
``ops.softmax_mse``."""
import argparse
import random

import numpy as np
import torch

from .. import config as cfg
from .. import ops, parallel
from ..network.unet import UNet
from .baseTrainer import seed_all, BaseTrainer, make_sgd


class meanTeacherTrainer(BaseTrainer):
    def __init__(self, phase, args=None):
        super().__init__(phase, args)
        self.lambda_semi = 1          # :41
        self.ema_decay = 0.99
        self.epoch_rampup = 30
        self.alpha = 0
        self.semi_start_iter = 100    # :123
        self.log_step = 50

    def build_network(self):
        self.net = UNet(cfg.img_channels, cfg.n_label + 1, cfg.base_width, norm_type="instance", act_type="lrelu").to(self.device)
        parallel.broadcast_parameters(self.net, self.group)
        if self.phase == "train":
            # the reference draws an independent random init for the teacher (:53) and never copies the student in
            self.ema = UNet(cfg.img_channels, cfg.n_label + 1, cfg.base_width, norm_type="instance", act_type="lrelu").to(self.device)
            for p in self.ema.parameters():
                p.requires_grad_(False)
            parallel.broadcast_parameters(self.ema, self.group)
            self.optimizer = make_sgd(self.net.parameters(), cfg.lr, 0.9, cfg.weight_decay)
            self.reducer = parallel.GradAllReducer(self.net.parameters(), self.group)

    def update_ema_variable(self):
        """:63-69 -- ema = alpha * ema + (1 - alpha) * student, alpha = 0 for the first 100 iterations."""
        self.alpha = 0 if self.iter < 100 else min(1 - 1 / (self.iter + 1), self.ema_decay)
        ema, cur = list(self.ema.parameters()), [p.detach() for p in self.net.parameters()]
        torch._foreach_mul_(ema, self.alpha)
        torch._foreach_add_(ema, cur, alpha=1 - self.alpha)

    def train_iteration(self, img, msk, noise=None):
        """One iteration of :86-149 on ``img`` = [labeled | unlabeled] (bs + bs slices).  Returns the device tensor
        [seg_loss, semi_loss]."""
        bs = msk.size(0)
        ul_img = img[bs:]
        if noise is None:
            noise = torch.clamp(torch.randn_like(ul_img) * 0.01, -0.02, 0.02)                  # :104
        lambda_semi = self.lambda_semi * self.sigmoid_rampup(self.epoch, self.epoch_rampup)
        with ops.wino_prepared(self.net, self.ema):         # (student: optimizer step; teacher: EMA update -- both after)
            out = self.net(img)
            with torch.no_grad():
                ema_out = self.ema(ul_img + noise)
            seg = self.loss(out[:bs], msk)
            if self.iter < self.semi_start_iter:
                semi = torch.zeros((), device=self.device)
            else:
                semi = ops.softmax_mse(out[bs:], ema_out)                                         # :129-131
            total = seg + lambda_semi * semi
            self.optimizer.zero_grad(set_to_none=True)
            total.backward()
        self.reducer.reduce()
        self.optimizer.step()
        self.update_ema_variable()
        lr_ = self.poly_lr()
        for g in self.optimizer.param_groups:
            g["lr"] = lr_
        self.iter += 1
        return torch.stack([seg.detach(), semi.detach()])

    def train_epoch(self, lb_loader, ul_loader, meter):
        self.net.train()
        lb_itr, ul_itr = iter(lb_loader), iter(ul_loader)
        for i in range(cfg.num_iter_per_epoch):
            try:
                img1, msk, mdl1, _ = next(lb_itr)
            except StopIteration:
                lb_itr = iter(lb_loader); img1, msk, mdl1, _ = next(lb_itr)
            try:
                img2, _, _, _ = next(ul_itr)
            except StopIteration:
                ul_itr = iter(ul_loader); img2, _, _, _ = next(ul_itr)
            img = torch.cat([img1.to(self.device, non_blocking=True), img2.to(self.device, non_blocking=True)], 0)
            scal = self.train_iteration(img, msk.to(self.device, non_blocking=True))
            if meter is not None:
                v, n = meter.collect_loss_by(scal[0].item(), mdl1[0].item(), img.size(0))
                meter.accumulate(v, n)
            if (i + 1) % self.log_step == 0:
                self.info("Iter %d, global_iter: %d, semi_loss: %.4f, seg_loss: %.4f, alpha: %f"
                          % (i, self.iter, scal[1].item(), scal[0].item(), self.alpha))


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-p", "--phase", type=str, choices=("train", "test"))
    p.add_argument("-f", "--fold", type=int, default=0)
    p.add_argument("-nm", "--expr_name", type=str)
    p.add_argument("-i", "--model_id", type=str)
    p.add_argument("-wh", "--which_ckpt", type=str, default="last")
    args = p.parse_args(argv)
    seed_all()
    t = meanTeacherTrainer(args.phase, args)
    if args.phase == "train":
        t.fit("inTurn")
    else:
        t.load_model(args.model_id, args.which_ckpt)
        t.test("inTurn", t.expr_root + "/" + args.model_id)


if __name__ == "__main__":
    main()
