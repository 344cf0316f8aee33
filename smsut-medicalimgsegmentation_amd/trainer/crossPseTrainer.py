"""``crossPseTrainer`` (reference trainer/crossPseTrainer.py:38-148): two U-Nets supervising each other with
argmax pseudo labels on the unlabeled half (cross pseudo supervision), DiceCE everywhere, two SGDs + poly LR.
Pure reuse of the U-Net kernels plus ``ops.argmax_channels``."""
import argparse
import random

import numpy as np
import torch

from .. import config as cfg
from .. import ops, parallel
from ..network.unet import UNet
from .baseTrainer import seed_all, BaseTrainer, make_sgd


class crossPseTrainer(BaseTrainer):
    def __init__(self, phase, args=None):
        super().__init__(phase, args)
        self.lambda_semi = 0.1        # :42
        self.log_step = 50

    def build_network(self):
        mk = lambda: UNet(cfg.img_channels, cfg.n_label + 1, cfg.base_width, norm_type="instance", act_type="lrelu").to(self.device)
        self.net, self.net2 = mk(), mk()
        parallel.broadcast_parameters(self.net, self.group)
        parallel.broadcast_parameters(self.net2, self.group)
        if self.phase == "train":
            self.optimizer1 = make_sgd(self.net.parameters(), cfg.lr, 0.9, cfg.weight_decay)
            self.optimizer2 = make_sgd(self.net2.parameters(), cfg.lr, 0.9, cfg.weight_decay)
            self.reducer1 = parallel.GradAllReducer(self.net.parameters(), self.group)
            self.reducer2 = parallel.GradAllReducer(self.net2.parameters(), self.group)

    def train_iteration(self, img, msk):
        """One iteration of :84-146; ``img`` = [labeled | unlabeled].  Returns [seg1, seg2, semi1, semi2] (device)."""
        bs = msk.size(0)
        lambda_semi = self.lambda_semi * self.sigmoid_rampup(self.epoch, cfg.max_epoch)        # :82
        with ops.wino_prepared(self.net, self.net2):        # (the weights move in the optimizer steps only)
            out1 = self.net(img)
            s1 = self.loss(out1[:bs], msk)
            out2 = self.net2(img)
            s2 = self.loss(out2[:bs], msk)
            pred1 = ops.argmax_channels(out1[bs:])                                                 # :122-125 (detached)
            pred2 = ops.argmax_channels(out2[bs:])
            semi1 = self.loss(out1[bs:], pred2)
            semi2 = self.loss(out2[bs:], pred1)
            total = s1 + s2 + lambda_semi * semi1 + lambda_semi * semi2
            self.optimizer1.zero_grad(set_to_none=True)
            self.optimizer2.zero_grad(set_to_none=True)
            total.backward()
        self.reducer1.reduce(); self.reducer2.reduce()
        self.optimizer1.step(); self.optimizer2.step()
        lr_ = self.poly_lr()
        for g in list(self.optimizer1.param_groups) + list(self.optimizer2.param_groups):
            g["lr"] = lr_
        self.iter += 1
        return torch.stack([t.detach() for t in (s1, s2, semi1, semi2)])

    def train_epoch(self, lb_loader, ul_loader, meter):
        self.net.train(); self.net2.train()
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
                vals = scal.tolist()
                for k in (0, 1):                                                              # :112-120: both nets are metered
                    v, n = meter.collect_loss_by(vals[k], mdl1[0].item(), img.size(0))
                    meter.accumulate(v, n)
            if (i + 1) % self.log_step == 0:
                self.info("Iter %d, global_iter: %d, crossPse1_loss: %.4f, crossPse2_loss: %.4f, seg1_loss: %.4f, seg2_loss: %.4f"
                          % (i, self.iter, scal[2].item(), scal[3].item(), scal[0].item(), scal[1].item()))


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-p", "--phase", type=str, choices=("train", "test"))
    p.add_argument("-f", "--fold", type=int, default=0)
    p.add_argument("-nm", "--expr_name", type=str)
    p.add_argument("-i", "--model_id", type=str)
    p.add_argument("-wh", "--which_ckpt", type=str, default="last")
    args = p.parse_args(argv)
    seed_all()
    t = crossPseTrainer(args.phase, args)
    if args.phase == "train":
        t.fit("inTurn")
    else:
        t.load_model(args.model_id, args.which_ckpt)
        t.test("inTurn", t.expr_root + "/" + args.model_id)


if __name__ == "__main__":
    main()
