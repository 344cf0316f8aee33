"""``UGANTrainer`` (reference trainer/uganTrainer.py:33-229): the fully supervised predecessor of the consistency
trainer -- ``UGAN`` (no PatchNCE head), every slice labeled, and a shape term DiceCE(seg(cycle), label) whose weight
ramps with the epoch (:122-123) in place of the consistency / NCE terms.  D-step identical to the hot loop."""
import argparse
import random

import numpy as np
import torch

from .. import config as cfg
from .. import ops, parallel
from ..network.ugan import UGAN, Discriminator
from .baseTrainer import seed_all, make_adam, make_sgd
from .uganShp0Trainer import UGANShp0Trainer

SCALARS = ("D_real", "D_fake", "D_cls", "D_gp", "G_fake", "G_rec", "G_cls", "G_seg", "G_shp")


class UGANTrainer(UGANShp0Trainer):
    def __init__(self, phase, args=None):
        self.lambda_shp, self.lambda_shp_lazy = 10, 20                                         # :40-41
        super().__init__(phase, args)

    def build_network(self):
        self.net = UGAN(cfg.img_channels, cfg.n_label + 1, cfg.n_modal, cfg.base_width).to(self.device)
        self.D = Discriminator(cfg.input_size, cfg.n_modal, cfg.base_width,
                               max_width=256 if cfg.base_width == 16 else 512).to(self.device)
        parallel.broadcast_parameters(self.net, self.group)
        parallel.broadcast_parameters(self.D, self.group)
        if self.phase == "train":
            self.optimizer = make_sgd(self.net.parameters(), cfg.lr, 0.9, cfg.weight_decay)
            self.d_optimizer = make_adam(self.D.parameters(), cfg.lr, [self.beta1, self.beta2], cfg.weight_decay)
            self.g_reducer = parallel.GradAllReducer(self.net.parameters(), self.group)
            self.d_reducer = parallel.GradAllReducer(self.D.parameters(), self.group)

    def _forward_eval(self, img):
        seg, _ = self.net(img)
        return seg

    def train_iteration(self, x_real, y_real, modal_org, mj=None, alpha=None):
        """One iteration of :134-222 (n_critic = 1); returns the 9 scalars of ``SCALARS`` as a device tensor."""
        lambda_shp = min(self.epoch * (self.lambda_shp / self.lambda_shp_lazy), self.lambda_seg)   # :122-123
        if mj is None:
            mj = random.randint(0, cfg.n_modal - 1)
        modal_org = modal_org.to(self.device).to(torch.int64)
        modal_trg = torch.full_like(modal_org, mj)
        vec_org = torch.zeros(modal_org.numel(), cfg.n_modal, device=self.device).scatter_(1, modal_org[:, None], 1.0)
        vec_trg = torch.zeros_like(vec_org)
        vec_trg[:, mj] = 1.0
        vec_ot, vec_to = vec_trg - vec_org, vec_org - vec_trg
        if alpha is None:
            alpha = torch.randn(x_real.size(0), 1, 1, 1, device=self.device)
        b = x_real.size(0)
        d_params = list(self.D.parameters())

        # G(x_real) once: the D-step uses it detached, the G-step continues from its graph (weights unchanged between)
        with ops.wino_prepared(self.net, forms="f"):
            y_fake, x_fake = self.net(x_real, vec_ot)

        # ---- D-step (:155-174)
        with ops.wino_prepared(self.D):
            with ops.first_order_pass():
                out_src, out_cls = self.D(torch.cat([x_real, x_fake.detach()], 0))
            d_real = ops.mean_all(out_src[:b], -1.0)
            d_cls = ops.cross_entropy_rows(out_cls[:b], modal_org)
            d_fake = ops.mean_all(out_src[b:], 1.0)
            x_hat = ops.row_lerp(x_real, x_fake.detach(), alpha).requires_grad_(True)
            out_src, _ = self.D(x_hat)                  # differentiated twice (gradient penalty): default op families
            d_gp = self.gradient_penalty(out_src, x_hat)
            d_loss = d_real + d_fake + self.lambda_cls * d_cls + self.lambda_gp * d_gp
            self.d_optimizer.zero_grad(set_to_none=True)
            d_loss.backward()
        self.d_reducer.reduce()
        self.d_optimizer.step()

        # ---- G-step (:178-198), D frozen
        for p in d_params:
            p.requires_grad_(False)
        with ops.wino_prepared(self.net), ops.wino_prepared(self.D):
            with ops.first_order_pass():
                out_src, out_cls = self.D(x_fake)
            g_fake = ops.mean_all(out_src, -1.0)
            g_cls = ops.cross_entropy_rows(out_cls, modal_trg)
            g_seg = self.loss(y_fake, y_real)
            y_rec, x_rec = self.net(x_fake, vec_to)
            g_rec = ops.l1_mean(x_real, x_rec)
            g_shp = self.loss(y_rec, y_real)
            g_loss = g_fake + self.lambda_rec * g_rec + self.lambda_cls * g_cls + self.lambda_seg * g_seg + lambda_shp * g_shp
            self.optimizer.zero_grad(set_to_none=True)
            g_loss.backward()
        for p in d_params:
            p.requires_grad_(True)
        self.g_reducer.reduce()
        self.optimizer.step()

        lr_ = self.poly_lr()
        for grp in list(self.optimizer.param_groups) + list(self.d_optimizer.param_groups):
            grp["lr"] = lr_
        self.iter += 1
        return torch.stack([t.detach().float() for t in (d_real, d_fake, d_cls, d_gp, g_fake, g_rec, g_cls, g_seg, g_shp)])

    def train_epoch(self, lb_loader, ul_loader, meter):
        self.net.train(); self.D.train()
        itr = iter(lb_loader)
        for i in range(self.n_critic * cfg.num_iter_per_epoch):
            try:
                x_real, y_real, modal_org, _ = next(itr)
            except StopIteration:
                itr = iter(lb_loader); x_real, y_real, modal_org, _ = next(itr)
            scal = self.train_iteration(x_real.to(self.device, non_blocking=True), y_real.to(self.device, non_blocking=True),
                                        modal_org)
            if meter is not None or (i + 1) % (self.n_critic * self.log_step) == 0:
                vals = scal.tolist()
                if meter is not None:
                    v, n = meter.collect_loss_by(vals[SCALARS.index("G_seg")], modal_org[0].item(), x_real.size(0))
                    meter.accumulate(v, n)
                if (i + 1) % (self.n_critic * self.log_step) == 0:
                    self.info("Iter: %d/%d(%d), " % (i, cfg.num_iter_per_epoch, self.iter)
                              + " ".join("%s: %.4f," % kv for kv in zip(SCALARS, vals)))


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-p", "--phase", type=str, choices=("train", "test"))
    p.add_argument("-f", "--fold", type=int, default=0)
    p.add_argument("-nm", "--expr_name", type=str)
    p.add_argument("-i", "--model_id", type=str)
    p.add_argument("-wh", "--which_ckpt", type=str, default="last")
    args = p.parse_args(argv)
    seed_all()
    t = UGANTrainer(args.phase, args)
    if args.phase == "train":
        t.fit("inTurn")
    else:
        t.load_model(args.model_id, args.which_ckpt)
        t.test("inTurn", t.expr_root + "/" + args.model_id)


if __name__ == "__main__":
    main()
