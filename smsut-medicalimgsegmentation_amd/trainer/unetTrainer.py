"""``UnetTrainer`` (reference trainer/unetTrainer.py:35-85): supervised U-Net, SGD(0.9, wd 1e-3), poly LR --
BASELINE configs 1-2."""
import argparse
import os
import random

import numpy as np
import torch

from .. import config as cfg
from .. import graphs, ops, parallel
from ..network.unet import UNet
from .baseTrainer import seed_all, BaseTrainer, make_sgd, sgd_step


class UnetTrainer(BaseTrainer):
    def build_network(self):
        self.net = UNet(cfg.img_channels, cfg.n_label + 1, cfg.base_width, norm_type="instance", act_type="lrelu")
        self.net.to(self.device)
        parallel.broadcast_parameters(self.net, self.group)
        if self.phase == "train":
            self.optimizer = make_sgd(self.net.parameters(), cfg.lr, 0.9, cfg.weight_decay)
            self.reducer = parallel.GradAllReducer(self.net.parameters(), self.group)
        self._graph = None

    # The step is two collective-free phases, split where data parallelism exchanges the Dice statistics
    # (batch_dice sums tp / fp / fn over the GLOBAL batch, misc/loss.py:52); each is captured as a hipGraph, the 16-float
    # all-reduce runs eagerly between the two replays (a no-op on one GPU).
    def _fwd_phase(self, img, msk):
        """forward + Dice/CE statistics; leaves the autograd graph for the second phase."""
        with ops.wino_prepared(self.net, forms="f"):        # the weights move in optimizer.step() only
            out = self.net(img)
        self._out = out
        return self.loss.stats(out, msk)

    def _bwd_phase(self, msk, stats):
        """loss from the (global) statistics + backward."""
        loss = self.loss.from_stats(self._out, msk, stats)
        with ops.wino_prepared(self.net, forms="b"):
            loss.backward()
        self._out = None
        return loss.detach()

    def graph_report(self):
        return {"mode": "graph" if self._graph else "eager", "captured": ["fwd", "bwd"] if self._graph else [], "fallback": False,
                "policy": os.environ.get("SMSUT_GRAPH", "default")}

    def train_step(self, img, msk):
        """One iteration of unetTrainer.py:66-83 (forward, DiceCE, zero_grad, backward, step, poly LR).
        Returns the loss as a 0-dim device tensor (no host sync)."""
        if graphs.graphs_enabled():
            key = tuple(img.shape)
            if self._graph is None or self._graph[0] != key:
                # a refused capture raises (no silent eager fallback; SMSUT_GRAPH=0 is the explicit eager mode)
                # warm-up outside any capture (first launches load code objects, the allocator sizes its pools): one
                # forward + backward whose gradients are dropped -- the weights do not move
                self._bwd_phase(msk, self._fwd_phase(img, msk))
                self.optimizer.zero_grad(set_to_none=True)
                torch.cuda.synchronize()
                ga = graphs.GraphedPhase(self._fwd_phase, (img, msk), [], warmup=0)
                self.loss.reduce_stats([ga.static_out])
                gb = graphs.GraphedPhase(self._bwd_phase, (msk, ga.static_out), self.net.parameters(), warmup=0)
                self._graph = (key, ga, gb)
                loss = gb.static_out                                     # the constructors already replayed them once
            else:
                stats = self._graph[1](img, msk)
                self.loss.reduce_stats([stats])
                loss = self._graph[2](msk, stats)
        else:
            if self._graph is not None:
                graphs.invalidate_grad_bindings()
            self.optimizer.zero_grad(set_to_none=True)
            stats = self._fwd_phase(img, msk)
            self.loss.reduce_stats([stats])
            loss = self._bwd_phase(msk, stats)
        self.reducer.reduce()
        sgd_step(self.optimizer)
        lr_ = self.poly_lr()
        for g in self.optimizer.param_groups:
            g["lr"] = lr_
        self.iter += 1
        return loss

    def train_epoch(self, lb_loader, ul_loader, meter):
        self.net.train()
        it = iter(lb_loader)
        from ..misc.utils import ScalarFetcher
        fetch = ScalarFetcher(1, self.device)                # (the loss reaches the meter one iteration late: no device stall)

        def consume(done):
            if done is not None and meter is not None:
                (val,), (modality, bsz) = done
                v, n = meter.collect_loss_by(val, modality, bsz)
                meter.accumulate(v, n)
        try:
            for _ in range(cfg.num_iter_per_epoch):
                try:
                    img, msk, mdl, _ = next(it)
                except StopIteration:
                    it = iter(lb_loader)
                    img, msk, mdl, _ = next(it)
                loss = self.train_step(img.to(self.device, non_blocking=True), msk.to(self.device, non_blocking=True))
                consume(fetch.push(loss.reshape(1), (int(mdl[0]), img.size(0))))
        finally:
            consume(fetch.flush())                           # (also when the loop raised: the last finished step still counts)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-p", "--phase", type=str, choices=("train", "test"))
    p.add_argument("-f", "--fold", type=int, default=0)
    p.add_argument("-nm", "--expr_name", type=str)
    p.add_argument("-i", "--model_id", type=str)
    p.add_argument("-wh", "--which_ckpt", type=str, default="last")
    args = p.parse_args(argv)
    seed_all()
    t = UnetTrainer(args.phase, args)
    if args.phase == "train":
        t.fit("inTurn")
    else:
        t.load_model(args.model_id, args.which_ckpt)
        t.test("inTurn", t.expr_root + "/" + args.model_id)


if __name__ == "__main__":
    main()
