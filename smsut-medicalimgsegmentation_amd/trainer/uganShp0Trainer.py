"""``UGANShp0Trainer`` pieces inherited by the consistency trainer (reference trainer/uganShp0Trainer.py:36-287):
``build_network`` (UGANnce + Discriminator + PatchNCE, SGD / Adam), ``load_model`` / ``save_model`` with the
``{prefix}_G.ckpt`` / ``{prefix}_D.ckpt`` names, ``label2onehot``, ``create_vectors``, ``denorm``,
``gradient_penalty`` and the ``val_phase`` validation forward."""
from os.path import join as pjoin

import numpy as np
import torch

from .. import config as cfg
from .. import ops, parallel
from ..network.patchnce import PatchNCELoss
from ..network.ugan import Discriminator, UGANnce
from .baseTrainer import BaseTrainer, make_adam, make_sgd


class UGANShp0Trainer(BaseTrainer):
    def __init__(self, phase, args=None):
        self.lambda_cls, self.lambda_rec, self.lambda_gp, self.lambda_seg = 1, 10, 10, 10     # :39-42
        self.log_step, self.n_critic = 50, 1
        self.beta1, self.beta2 = 0.9, 0.999
        super().__init__(phase, args)

    def build_network(self):
        self.net = UGANnce(cfg.img_channels, cfg.n_label + 1, cfg.n_modal, cfg.base_width).to(self.device)
        self.criterionNCE = [PatchNCELoss(cfg.batch_size) for _ in cfg.nce_layers]                 # :57-59
        self.D = Discriminator(cfg.input_size, cfg.n_modal, cfg.base_width,
                               max_width=256 if cfg.base_width == 16 else 512).to(self.device)     # :61-63
        parallel.broadcast_parameters(self.net, self.group)
        parallel.broadcast_parameters(self.D, self.group)
        if self.phase == "train":
            self.optimizer = make_sgd(self.net.parameters(), cfg.lr, 0.9, cfg.weight_decay)
            self.d_optimizer = make_adam(self.D.parameters(), cfg.lr, [self.beta1, self.beta2], cfg.weight_decay)
            self.g_reducer = parallel.GradAllReducer(self.net.parameters(), self.group)
            self.d_reducer = parallel.GradAllReducer(self.D.parameters(), self.group)

    def load_model(self, model_idx, which_ckpt):
        root = pjoin(self.expr_root, model_idx, "ckpt")
        self.net.load_state_dict(torch.load(pjoin(root, f"{which_ckpt}_G.ckpt"), map_location="cpu"))
        self.D.load_state_dict(torch.load(pjoin(root, f"{which_ckpt}_D.ckpt"), map_location="cpu"))
        self.net.to(self.device); self.D.to(self.device)
        self.info(f"[*] Load G and D from {root}.")

    def save_model(self, prefix):
        if self.rank == 0:
            root = pjoin(self.expr_root, self.model_idx, "ckpt")
            # .contiguous(): checkpoints hold plain OIHW tensors, loadable by the reference's nn.Modules
            torch.save({k: v.contiguous() for k, v in self.net.state_dict().items()}, pjoin(root, f"{prefix}_G.ckpt"))
            torch.save({k: v.contiguous() for k, v in self.D.state_dict().items()}, pjoin(root, f"{prefix}_D.ckpt"))
            self.info(f"[*] Save G and D to {root}.")
        self.save_train_state(prefix)                       # collective under DP: every rank's RNG / loader state

    def label2onehot(self, modals, dim=cfg.n_modal):
        out = torch.zeros(modals.size(0), dim)
        out[np.arange(modals.size(0)), modals.long().cpu()] = 1
        return out

    def create_vectors(self, vec_org, dim):
        return [self.label2onehot(torch.ones(vec_org.size(0)) * i, dim).to(self.device) for i in range(dim)]

    @staticmethod
    def denorm(x):
        return ((x + 1.0) / 2.0).clamp_(0, 1)

    def gradient_penalty(self, y, x):
        """:127-134 -- ``create_graph=True`` so ``d_loss.backward()`` differentiates D's backward again.  The
        ``input_grads_only`` scope tells the HIP backward ops that only d/dx is wanted here."""
        weight = torch.ones(y.size(), device=self.device)
        with ops.input_grads_only():
            dydx = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=weight, retain_graph=True, create_graph=True,
                                       only_inputs=True)[0]
        return ops.grad_penalty(dydx)

    def _forward_eval(self, img):
        seg, _ = self.net(img, val_phase=True)            # :267
        return seg
