"""``UGANConsisTrainer`` -- the hot loop the north star names (reference trainer/uganConsisTrainer.py:35-214).

``train_iteration`` is one pass of :110-203: D-step (D(real), G(real->fake) detached, D(fake), WGAN-GP on x_hat,
Adam) then G-step (G, D(fake), DiceCE on the labeled half, cycle reconstruction G(fake->rec), L1, consistency
DiceCE against argmax pseudo labels from iteration 1000 on, PatchNCE, SGD), then the poly LR write.

MI355X-first choices that do not change the arithmetic:
  * the D-step's generator forward runs under ``no_grad`` (the reference builds a graph and then ``.detach()``es);
  * D's parameters are frozen during the G-step, so the unused D gradients the reference's ``g_loss.backward()``
    fills (and never steps) are neither computed nor all-reduced;
  * the 10 logged scalars stay on the device and are fetched with ONE host sync when a caller asks for them
    (the reference does 11 ``.item()`` syncs per iteration);
  * RNG-dependent inputs (target modality, alpha ~ randn, patch ids ~ randperm) may be passed in for replay.
"""
import argparse
import os
import random
import time
from os.path import join as pjoin

import numpy as np
import torch

from .. import config as cfg
from .. import graphs, ops
from .baseTrainer import seed_all, sgd_step
from .uganShp0Trainer import UGANShp0Trainer

SCALARS = ("D_real", "D_fake", "D_cls", "D_gp", "G_fake", "G_rec", "G_cls", "G_seg", "G_semi", "G_nce")


class UGANConsisTrainer(UGANShp0Trainer):
    def __init__(self, phase, args=None):
        super().__init__(phase, args)
        self.lambda_semi = 10
        self.semi_start_iter = 1000                      # :165
        self._semi_on = False
        self._graphs = {}
        self._alias = None
        self._d_alias = None
        self._use_d_alias = os.environ.get("SMSUT_D_ALIAS", "1") not in ("0", "")
        self._d_async = os.environ.get("SMSUT_D_ASYNC_ALLREDUCE", "0") not in ("0", "")
        self._g1 = self._g2 = self._seg_real = None
        self._side = None
        # SMSUT_D_OVERLAP = 1: D-step (graph, gradient all-reduce, Adam) and D(x_fake) on a side stream beside the generator's
        # cycle pass and D-independent backward -- the default at ONE GPU.  Under data parallelism (world > 1, any backend) that
        # variant would issue collectives from two streams; it has only ever run at one rank over RCCL and with two gloo ranks, so
        # it stays an A/B switch (``bench.py --d-overlap 1``).  The data-parallel default is 2 (below): the side stream carries
        # captured COMPUTE only and every collective stays on the main stream.  0: everything on one stream.
        ov = os.environ.get("SMSUT_D_OVERLAP", "") or ("1" if self.world == 1 else "2")
        if ov not in ("0", "1", "2"):        # ('true' / '3' used to select the one-stream schedule silently)
            raise ValueError(f"SMSUT_D_OVERLAP must be 0, 1 or 2 (got {ov!r})")
        self._d_overlap = ov == "1"
        # SMSUT_D_OVERLAP=2 -- "compute-only side stream": only the captured D-step (collective-free) runs on the side stream,
        # beside G2gen AND G2a; it is JOINED before its gradient all-reduce, so every collective (statistics, D gradients, G
        # gradients) is issued from the main stream in program order -- for RCCL this is the one-stream configuration; Adam and
        # D(x_fake) follow on the main stream.
        self._d_side_compute = ov == "2"
        # SMSUT_SEG_BATCH (r05, default on): the SEGMENTATION branch of the two generator passes of an iteration -- seg_encoder, enc5,
        # seg_decoder on x_real (uganConsisTrainer.py:152) and on x_fake (the cycle pass, :159) -- runs as ONE pass over the batch
        # [x_real | x_fake]: the branch does not depend on the translation branch of its own pass, only on its input, and InstanceNorm
        # is per sample, so every value is the one the two passes produce -- but every kernel of that half of the generator runs
        # once on 32 slices instead of twice on 16 (forward, data- and weight-gradient).  The translation branches stay two passes
        # (the second one reads the first one's output); their 3x3 weight gradients are paired (ops.pair_wgrads).
        self._seg_batch = os.environ.get("SMSUT_SEG_BATCH", "1") not in ("0", "")
        # SMSUT_D_EARLY_ALLREDUCE=1 (mode 2 only): D's gradient all-reduce is STARTED (pack on the side stream, collective on RCCL's own)
        # before G2a1 and collected after it.  Off by default: at ONE rank it measured slower (21.25 vs 20.85 ms per iteration over a
        # one-rank RCCL communicator, profiles/r05_notes.md) -- there is no wire time to hide there, only one more stream to
        # synchronise with; an A/B switch for the first multi-GPU run, where the 9.7 MB ring all-reduce is what it would hide.
        self._d_early = os.environ.get("SMSUT_D_EARLY_ALLREDUCE", "0") not in ("0", "")
        # The G-step's backward in THREE pieces (SMSUT_G_SPLIT; default: on whenever a side stream is in use), so that only what
        # needs the updated D waits for it:
        #   G2a  backward of the D-independent terms (cycle L1, PatchNCE, both DiceCE values) -- the whole cycle pass and the
        #        segmentation branch of G(x_real), ~70 % of the generator's backward -- stopping at two cut points of G(x_real)'s
        #        graph: x_fake and the bottleneck features t_e5 (the cycle pass and netF read DETACHED copies of them);
        #   G2d  D(x_fake) through the updated D, forward and data-gradient: d(g_fake + lambda_cls g_cls) / d x_fake;
        #   G2c  the translation branch of G(x_real) from the summed cut-point gradients.
        # Same sums as one g_loss.backward() (autograd adds the same contributions at x_fake / t_e5; only the order in which a
        # shared parameter's gradient is accumulated differs, 1e-7).  The D-step's ~700 small launches then hide under 12 ms of
        # chip-filling generator kernels instead of 2.8 (one GPU: 23.3 -> 21.9 ms per iteration).  On ONE stream the pieces run
        # back to back and measure 0.5 % slower than one backward (24.51 vs 24.38 ms): off there.
        self._g_split = os.environ.get("SMSUT_G_SPLIT", "1" if (self._d_overlap or self._d_side_compute) else "0") not in ("0", "")
        # Forks INSIDE captured phases: the D-step's twice-differentiated x_hat pass (forward and -- autograd replays a node on its
        # forward's stream -- both of its backward sweeps) beside the batched real | fake pass, and D(x_fake) of the one-piece
        # G-step beside the segmentation branch's backward.  They shorten the D-step when it is on the critical path (one-piece
        # G-step with the side stream: 24.6 -> 23.5 ms per iteration) -- and COST 2 % once the three-piece G-step has taken it
        # off that path (22.0 with, 21.57 without: the D chain has slack there, and forking it only adds contention for the
        # generator's kernels); on one stream they are neutral.  Default: on only in the configuration they help.
        fork_default = "1" if (self.world == 1 and self._d_overlap and not self._g_split) else "0"
        self._d_fork = os.environ.get("SMSUT_D_FORK", fork_default) not in ("0", "")
        self._fork = None
        self._g2_fork = os.environ.get("SMSUT_G2_FORK", fork_default) not in ("0", "")
        self._eager_done = False
        self._lambda_semi_t = torch.zeros((), device=self.device)
        self._probe = os.environ.get("SMSUT_DEBUG_FINITE", "0") not in ("0", "")
        self.finite_log = []

    def consistency_loss(self, source, target):
        return self.loss(source, ops.argmax_channels(target))                    # :45-53

    def nce_loss(self, feat_x_pool, feat_f_pool):
        total = 0.0
        for f_f, f_x, crit in zip(feat_f_pool, feat_x_pool, self.criterionNCE):     # :55-64
            total = total + ops.mean_all(crit(f_f, f_x), 1.0)
        return total / len(cfg.nce_layers)

    def _keep_stats(self, slot, st):
        """The Dice statistics of the two loss terms live back to back in ONE persistent buffer (slot 0: segmentation term, phase
        G1; slot 1: consistency term, phase G2gen): under data parallelism the exchange between the phases is then a single
        in-place all-reduce of that buffer -- no pack, no unpack on the critical path (ops.all_reduce_dice_stats)."""
        k = st.numel()
        buf = self.__dict__.get("_sflat")
        if buf is None or buf.numel() != 2 * k or buf.device != st.device:
            buf = self._sflat = torch.zeros(2 * k, dtype=torch.float32, device=st.device)
        view = buf[slot * k:(slot + 1) * k]
        view.copy_(st)
        return view

    def _finite_probe(self, tag, named):
        """``SMSUT_DEBUG_FINITE=1``: after each phase, record the first non-finite tensor (scalars, gradients, parameters,
        translated images) together with the iteration and the RNG draws that led to it.  One host sync per probe --
        a diagnosis mode, never on in timed runs."""
        for name, t in named:
            if t is None:
                continue
            if not bool(torch.isfinite(t).all()):
                self.finite_log.append((self.iter, tag, name))
                return False
        return True

    def _to_device_async(self, t):
        if t.is_cuda:
            return t.to(torch.int64)
        # a fresh pinned tensor per call: the caching host allocator records the copy's stream event and does not hand
        # the block out again before the copy has run (one shared staging buffer could be overwritten by the host for
        # iteration k+1 while iteration k's copy was still queued behind ~30 ms of GPU work)
        pin = torch.empty(t.numel(), dtype=torch.int64, pin_memory=True)
        pin.copy_(t.reshape(-1))
        return pin.to(self.device, non_blocking=True)

    # ------------------------------------------------------------------ the four phases of one iteration
    # The reference runs G(x_real) twice per iteration with the SAME generator weights: once in the D-step (detached,
    # :133-135) and again in the G-step (:151) -- D is updated in between, G is not, so both calls produce identical
    # values.  Here it is computed ONCE (phase G1, with its autograd graph kept), the D-step consumes the detached
    # result, and the G-step continues from the stored graph.  D(x_real) and D(x_fake) of the D-step share one batched
    # pass (InstanceNorm is per sample, so the values are unchanged).
    #
    # Phase boundaries sit exactly where data parallelism needs a collective, so every phase is collective-free and is
    # captured as a hipGraph on one GPU and under DP alike (r01 ran G1 / G2 eagerly under DP):
    #   G1   G(x_real)                      + Dice statistics of (y_fake[:bs], y_real)
    #   D    D-step forward / backward      -> [all-reduce of D's gradients + Adam: side stream, overlapped with G2gen]
    #   G2gen  cycle pass G(x_fake), L1, PatchNCE, pseudo labels + Dice statistics of (y_rec, argmax y_fake)
    #                                       -> [ONE all-reduce of both statistics sets]
    #   G2   D(x_fake) through the UPDATED frozen D, both DiceCE values from the global statistics, g_loss.backward()
    def _g1_phase(self, x_real, vec_ot, ids, y_real):
        """G(x_real -> x_fake) with the autograd graph retained for the G-step; returns (x_fake detached, seg statistics)."""
        # (scopes: the weights move in the optimizer steps only; pair_wgrads: this pass and the cycle pass of phase G2gen go through
        #  every generator layer with the same weights -- their 3x3 weight gradients are computed by ONE launch per layer, ops.py)
        if self._seg_batch:
            # translation branch only; the segmentation branch of x_real runs batched with the cycle pass' in phase G2gen
            with ops.wino_prepared(self.net, forms="f"), ops.pair_wgrads():
                if self._g_split:
                    x_fake, _, _, feat_x = self.net(x_real, vec_ot, branch="tsl")             # feat_x: the raw bottleneck features t_e5
                else:
                    x_fake, feat_x, _, _ = self.net(x_real, vec_ot, sample_ids=[ids], branch="tsl")
            self._g1 = (None, x_fake, feat_x)
            return x_fake.detach(), torch.zeros(1, device=self.device)
        with ops.wino_prepared(self.net, forms="f"), ops.pair_wgrads():
            if self._g_split:
                y_fake, x_fake, feat_x = self.net._trunk(x_real, vec_ot)          # feat_x: the raw bottleneck features t_e5
            else:
                y_fake, x_fake, feat_x, _ = self.net(x_real, vec_ot, sample_ids=[ids])
        self._g1 = (y_fake, x_fake, feat_x)
        return x_fake.detach(), self._keep_stats(0, self.loss.stats(y_fake[:y_real.size(0)], y_real))

    def _d_phase(self, x_real, x_fake, modal_org, alpha):
        """D-step forward + backward (:129-144).  Returns [D_real, D_fake, D_cls, D_gp]."""
        with ops.wino_prepared(self.D):
            return self._d_phase_body(x_real, x_fake, modal_org, alpha)

    def _d_phase_body(self, x_real, x_fake, modal_org, alpha):
        b = x_real.size(0)

        def xhat_pass():
            x_hat = ops.row_lerp(x_real, x_fake, alpha).requires_grad_(True)
            if self._d_alias is not None:
                # the x_hat pass on parameter ALIASES (same storage, separate .grad): every D parameter is reached by both passes,
                # and autograd would sum their gradients with one tiny add kernel per parameter; now one multi-tensor add (as for G)
                out_hat, _ = torch.func.functional_call(self.D, self._d_alias, (x_hat,))
            else:
                out_hat, _ = self.D(x_hat)          # differentiated twice (gradient penalty): default op families
            return self.gradient_penalty(out_hat, x_hat)

        if self._d_fork and self._d_alias is not None:
            cur = torch.cuda.current_stream()
            if self._fork is None:
                self._fork = torch.cuda.Stream()
            self._fork.wait_stream(cur)              # (inside a capture this brings the fork stream into it)
            with torch.cuda.stream(self._fork):
                gp = xhat_pass()

            def join():
                cur.wait_stream(self._fork)
                return gp
        else:
            join = xhat_pass
        with ops.first_order_pass():
            out_src, out_cls = self.D(torch.cat([x_real, x_fake], 0))
        d_real = ops.mean_all(out_src[:b], -1.0)
        d_cls = ops.cross_entropy_rows(out_cls[:b], modal_org)
        d_fake = ops.mean_all(out_src[b:], 1.0)
        d_gp = join()
        d_loss = d_real + d_fake + self.lambda_cls * d_cls + self.lambda_gp * d_gp
        d_loss.backward()
        if self._d_alias is not None:
            main, extra = [], []
            for name, p in self.D.named_parameters():
                a = self._d_alias[name]
                if a.grad is not None:
                    if p.grad is None:
                        p.grad = a.grad
                    else:
                        main.append(p.grad); extra.append(a.grad)
            if main:
                torch._foreach_add_(main, extra)
        return torch.stack([t.detach().float() for t in (d_real, d_fake, d_cls, d_gp)])

    def _g2gen_phase(self, x_real, vec_to, ids, y_real=None):
        """The cycle pass of the G-step (:159-168) -- everything that does not need the updated D: G(x_fake -> x_rec) on
        parameter ALIASES (same storage, separate ``.grad``, so the two passes' weight gradients are summed by one
        multi-tensor add instead of one add kernel per parameter), L1, PatchNCE, the pseudo labels and the consistency
        term's Dice statistics.  Leaves its autograd graph for phase G2; returns the (local) statistics.
        SMSUT_SEG_BATCH: the segmentation branch of BOTH passes runs here, once, on [x_real | x_fake] (main parameters), and the
        cycle pass on the aliases is its translation branch only; returns both statistics sets (one buffer: [seg | semi])."""
        y_fake, x_fake, feat_x = self._g1
        cut = None
        if self._g_split:
            # cut points of G(x_real)'s graph: this phase and G2a differentiate down to these leaves, G2c carries on from them
            xd, td = x_fake.detach().requires_grad_(True), feat_x.detach().requires_grad_(True)
            feat_x, _ = self.net.netF([td], patch_ids=[ids])
            cut, x_fake = (xd, td), xd
        if self._seg_batch:
            b, bs = x_real.size(0), y_real.size(0)
            with ops.wino_prepared(self.net, forms="f"):
                y_all = self.net(torch.cat([x_real, x_fake], 0), branch="seg")
                with ops.pair_wgrads():
                    x_rec, feat_f, _, _ = torch.func.functional_call(self.net, self._alias, (x_fake, vec_to),
                                                                     {"sample_ids": [ids], "branch": "tsl"})
            # the rows the losses read: the labeled half of G(x_real)'s segmentation, the whole of the cycle pass' (one gradient buffer)
            y_seg, y_rec = ops.row_segments(y_all, 0, bs, b, 2 * b)
            self._g1 = (y_seg, self._g1[1], self._g1[2])
            st_seg = self._keep_stats(0, self.loss.stats(y_seg, y_real))
            y_fake = y_all[:b].detach()
            self._seg_real = y_fake                  # (segmentation logits of G(x_real), all rows: read by tests / tools only)
        else:
            with ops.wino_prepared(self.net, forms="f"), ops.pair_wgrads():
                y_rec, x_rec, feat_f, _ = torch.func.functional_call(self.net, self._alias, (x_fake, vec_to),
                                                                     {"sample_ids": [ids]})
        g_rec = ops.l1_mean(x_real, x_rec)
        g_nce = self.nce_loss(feat_x, feat_f)
        if self._semi_on:
            pseudo = ops.argmax_channels(y_fake)                                     # :45-53
            st = self._keep_stats(1, self.loss.stats(y_rec, pseudo))
        else:
            pseudo, st = None, torch.zeros(1, device=self.device)
        self._g2 = (y_rec, pseudo, g_rec, g_nce) + ((cut,) if cut is not None else ())
        if self._seg_batch:
            return self._sflat                       # [seg statistics | consistency statistics]: back to back, reduced in place
        return st

    def _g2a_phase(self, y_real, st_seg, st_semi, lambda_semi):
        """SMSUT_G_SPLIT: losses that do not involve D and their backward, down to the cut points.  Returns
        [G_rec, G_seg, G_semi, G_nce]."""
        bs = y_real.size(0)
        y_fake = self._g1[0]
        y_rec, pseudo, g_rec, g_nce, _ = self._g2
        g_seg = self.loss.from_stats(y_fake if y_fake.size(0) == bs else y_fake[:bs], y_real, st_seg)
        if self._semi_on:
            g_semi = self.loss.from_stats(y_rec, pseudo, st_semi)
        else:
            g_semi = torch.zeros((), device=self.device)
        with ops.wino_prepared(self.net, forms="b"):
            (self.lambda_rec * g_rec + self.lambda_seg * g_seg + lambda_semi * g_semi + 1.0 * g_nce).backward()
        return torch.stack([t.detach().float() for t in (g_rec, g_seg, g_semi, g_nce)])

    # Mode 2 (data parallelism) cuts G2a once more, so that D(x_fake) -- which there must follow D's all-reduce and Adam on the
    # main stream's time line -- still has generator work to run beside: G2a1 = the cycle pass' terms, G2a2 = the segmentation
    # term of G(x_real) (disjoint graphs and parameters: two backward calls, same values).
    def _g2a1_phase(self, st_semi, lambda_semi):
        """[G_rec, G_semi, G_nce] and their backward (cycle pass on the aliases, netF, down to the cut points).
        SMSUT_SEG_BATCH: the segmentation branch is ONE graph over both passes, so its two terms go together (phase G2a2) and this
        phase holds the cycle pass' translation branch only: [G_rec, G_nce]."""
        y_rec, pseudo, g_rec, g_nce, _ = self._g2
        if self._seg_batch:
            with ops.wino_prepared(self.net, forms="b"):
                (self.lambda_rec * g_rec + 1.0 * g_nce).backward()
            return torch.stack([t.detach().float() for t in (g_rec, g_nce)])
        if self._semi_on:
            g_semi = self.loss.from_stats(y_rec, pseudo, st_semi)
        else:
            g_semi = torch.zeros((), device=self.device)
        with ops.wino_prepared(self.net, forms="b"):
            (self.lambda_rec * g_rec + lambda_semi * g_semi + 1.0 * g_nce).backward()
        return torch.stack([t.detach().float() for t in (g_rec, g_semi, g_nce)])

    def _g2a2_phase(self, y_real, st_seg, st_semi=None, lambda_semi=None):
        """[G_seg] and its backward (segmentation branch of G(x_real)).  SMSUT_SEG_BATCH: [G_seg, G_semi] -- both terms of the batched
        segmentation branch, one backward through it."""
        y_fake = self._g1[0]
        g_seg = self.loss.from_stats(y_fake if y_fake.size(0) == y_real.size(0) else y_fake[:y_real.size(0)], y_real, st_seg)
        if self._seg_batch:
            y_rec, pseudo = self._g2[0], self._g2[1]
            g_semi = self.loss.from_stats(y_rec, pseudo, st_semi) if self._semi_on else torch.zeros((), device=self.device)
            with ops.wino_prepared(self.net, forms="b"):
                (self.lambda_seg * g_seg + lambda_semi * g_semi).backward()
            return torch.stack([t.detach().float() for t in (g_seg, g_semi)])
        with ops.wino_prepared(self.net, forms="b"):
            (self.lambda_seg * g_seg).backward()
        return g_seg.detach().float().reshape(1)

    def _g2d_phase(self, modal_trg):
        """SMSUT_G_SPLIT: D(x_fake) through the updated, frozen D and its data-gradient.  Returns [G_fake, G_cls]; the
        gradient w.r.t. x_fake stays in ``self._gx_d`` for G2c."""
        xd2 = self._g1[1].detach().requires_grad_(True)
        with ops.wino_prepared(self.D):
            with ops.first_order_pass():
                out_src, out_cls = self.D(xd2)
            g_fake = ops.mean_all(out_src, -1.0)
            g_cls = ops.cross_entropy_rows(out_cls, modal_trg)
            (g_fake + self.lambda_cls * g_cls).backward()
        self._gx_d = xd2.grad
        return torch.stack([t.detach().float() for t in (g_fake, g_cls)])

    def _g2c_phase(self):
        """SMSUT_G_SPLIT: the translation branch of G(x_real) from the gradients summed at the cut points, then the aliases'
        gradients into the parameters'.  Returns a dummy (phases return a tensor)."""
        _, x_fake, t_e5 = self._g1
        xd, td = self._g2[4]
        with ops.wino_prepared(self.net, forms="b"):
            torch.autograd.backward([x_fake, t_e5], [xd.grad + self._gx_d, td.grad])
            ops.pair_flush()                     # sets parked without a partner (consistency term off): computed alone, here
        self._g1 = self._g2 = self._gx_d = self._seg_real = None
        main, extra = [], []
        for name, p in self.net.named_parameters():
            a = self._alias[name]
            if a.grad is not None:
                if p.grad is None:
                    p.grad = a.grad
                else:
                    main.append(p.grad); extra.append(a.grad)
        if main:
            torch._foreach_add_(main, extra)
        return torch.zeros(1, device=self.device)

    def _g2_phase(self, y_real, modal_trg, st_seg, st_semi, lambda_semi):
        """Rest of the G-step (:152-179) with D frozen: D(x_fake), the losses, backward through both generator passes.
        ``lambda_semi`` is a 0-dim device tensor (it changes every epoch and must not be baked into a captured graph);
        ``st_*`` are the Dice statistics, already summed over the ranks.
        Returns [G_fake, G_rec, G_cls, G_seg, G_semi, G_nce]."""
        with ops.wino_prepared(self.net, forms="b"), ops.wino_prepared(self.D):
            return self._g2_phase_body(y_real, modal_trg, st_seg, st_semi, lambda_semi)

    def _g2_phase_body(self, y_real, modal_trg, st_seg, st_semi, lambda_semi):
        bs = y_real.size(0)
        y_fake, x_fake, _ = self._g1
        y_rec, pseudo, g_rec, g_nce = self._g2
        def d_pass():
            with ops.first_order_pass():
                out_src, out_cls = self.D(x_fake)
            return ops.mean_all(out_src, -1.0), ops.cross_entropy_rows(out_cls, modal_trg)

        fork = self._d_fork and self._g2_fork
        if fork:
            # D(x_fake) on the fork stream: its backward -- ~200 small launches through the frozen D, the first nodes autograd
            # runs (they were recorded last) -- then goes out on that stream too, beside the segmentation branch's backward
            # (independent of D) instead of in front of it
            cur = torch.cuda.current_stream()
            if self._fork is None:
                self._fork = torch.cuda.Stream()
            self._fork.wait_stream(cur)
            with torch.cuda.stream(self._fork):
                g_fake, g_cls = d_pass()
        else:
            g_fake, g_cls = d_pass()
        g_seg = self.loss.from_stats(y_fake if y_fake.size(0) == bs else y_fake[:bs], y_real, st_seg)
        if self._semi_on:
            g_semi = self.loss.from_stats(y_rec, pseudo, st_semi)
        else:
            g_semi = torch.zeros((), device=self.device)
        if fork:
            cur.wait_stream(self._fork)
        g_loss = g_fake + self.lambda_rec * g_rec + self.lambda_cls * g_cls + self.lambda_seg * g_seg \
            + lambda_semi * g_semi + 1.0 * g_nce
        g_loss.backward()
        ops.pair_flush()                         # sets parked without a partner (consistency term off): computed alone, here
        self._g1 = self._g2 = None
        main, extra = [], []
        for name, p in self.net.named_parameters():
            a = self._alias[name]
            if a.grad is not None:
                if p.grad is None:
                    p.grad = a.grad
                else:
                    main.append(p.grad); extra.append(a.grad)
        if main:
            torch._foreach_add_(main, extra)
        return torch.stack([t.detach().float() for t in (g_fake, g_rec, g_cls, g_seg, g_semi, g_nce)])

    def _run_phase(self, name, fn, inputs, params, rebind=()):
        """Eager call, or capture-once / replay as a hipGraph (graphs.GraphedPhase) when enabled.  The very first
        iteration always runs eagerly (it is the warm-up the capture needs).  ``rebind``: parameters whose ``.grad`` this
        phase's backward fills although another phase clears them (the generator aliases in G2)."""
        use_graph = graphs.graphs_enabled() and self._eager_done
        if not use_graph:
            if self._graphs:
                graphs.invalidate_grad_bindings()          # an eager step re-points .grad: captured phases rebind on replay
            for p in params:
                p.grad = None
            return fn(*inputs)
        # (the split G-step phases take their operands through hidden state -- self._g1 / _g2 / _gx_d -- so the
        #  shapes of the explicit inputs do not identify the batch; the iteration's geometry is part of every key, else a
        #  second batch / image size on the same trainer would replay G2a1 / G2a2 / G2d / G2c against the OLD pools)
        key = (name, self._semi_on, self._geom) + tuple(tuple(t.shape) for t in inputs)
        g = self._graphs.get(key)
        if g is None:
            # A refused capture raises (r01 swallowed it and ran eagerly forever: after a partial capture p.grad points
            # into an aborted pool and G1's autograd graph may already be consumed -- there is nothing safe to fall back
            # to, and a silent eager run is a performance cliff nobody sees).  SMSUT_GRAPH=0 is the explicit eager mode.
            g = self._graphs[key] = graphs.GraphedPhase(fn, inputs, params, warmup=0, rebind_params=rebind)
            return g.static_out                                          # the capture pass does not execute: replay it
        return g(*inputs)

    def graph_report(self):
        """What actually ran: which phases are captured hipGraphs (bench.py prints this next to the timing)."""
        captured = sorted({k[0] for k in self._graphs if isinstance(k, tuple)})
        whole = set(captured) in ({"D", "G1", "G2", "G2gen"}, {"D", "G1", "G2a", "G2c", "G2d", "G2gen"},
                                  {"D", "G1", "G2a1", "G2a2", "G2c", "G2d", "G2gen"})      # (G-step: one phase, three, or four)
        mode = "graph" if whole else ("eager" if not captured else "graph(" + ",".join(captured) + ")")
        return {"mode": mode, "captured": captured, "fallback": False,
                "policy": os.environ.get("SMSUT_GRAPH", "default")}

    def train_iteration(self, x_real, y_real, modal_org, mj=None, alpha=None, sample_ids=None):
        """One iteration; returns a float32 device tensor with the 10 scalars in ``SCALARS`` order."""
        lambda_semi = self.lambda_semi * self.sigmoid_rampup(self.epoch, cfg.max_epoch)       # :74
        self._geom = (tuple(x_real.shape), tuple(y_real.shape))
        ops.pair_reset()                       # (paired weight gradients: nothing parked by an iteration that died half-way survives)
        if mj is None:
            mj = random.randint(0, cfg.n_modal - 1)                                           # :114
        # modality ids go to the device through a pinned staging buffer (a pageable .to(device) would make the host
        # wait for the whole launch queue every iteration); the one-hots (:116-117) are built on the device
        modal_org = self._to_device_async(modal_org)
        modal_trg = torch.full_like(modal_org, mj)
        vec_org = torch.zeros(modal_org.numel(), cfg.n_modal, device=self.device).scatter_(1, modal_org[:, None], 1.0)
        vec_trg = torch.zeros_like(vec_org)
        vec_trg[:, mj] = 1.0
        vec_ot, vec_to = vec_trg - vec_org, vec_org - vec_trg
        if alpha is None:
            alpha = torch.randn(x_real.size(0), 1, 1, 1, device=self.device)                  # randn, not rand (:138)
        if sample_ids is None:
            # PatchSampleF draws torch.randperm(H*W)[:64] on the bottleneck map (ugan.py:321-323): H/16 x W/16.  The
            # reference draws once in the D-step (unused there: x_fake is detached) and once in the G-step.
            hw = (x_real.shape[2] // 16) * (x_real.shape[3] // 16)
            torch.randperm(hw, device=self.device)                                             # the D-step's draw
            ids = torch.randperm(hw, device=self.device)[: min(64, hw)]
        else:
            ids = sample_ids[0].to(self.device)
        self._semi_on = self.iter >= self.semi_start_iter                                      # :165
        lam_t = self._lambda_semi_t.fill_(lambda_semi)
        if self._alias is None:
            self._alias = {k: p.detach().requires_grad_(True) for k, p in self.net.named_parameters()}
        if self._use_d_alias and self._d_alias is None:
            self._d_alias = {k: p.detach().requires_grad_(True) for k, p in self.D.named_parameters()}
        g_params = list(self.net.parameters())
        d_params = list(self.D.parameters()) + (list(self._d_alias.values()) if self._d_alias else [])

        # ------------------------------------------------------------ G(x_real): once, shared by both steps
        x_fake, st_seg = self._run_phase("G1", self._g1_phase, (x_real, vec_ot, ids, y_real), [])
        if self._probe:
            self._finite_probe("G1", [("x_fake", x_fake)])

        # ------------------------------------------------------------ D-step (:129-146)  ||  cycle pass of the G-step
        # The D-step (forward x3, WGAN-GP double backward, gradient all-reduce, Adam) and the cycle pass G(x_fake) of the
        # G-step are independent: both only READ x_fake.  The D-step is ~700 small launches (its deep levels are 8x8 / 4x4
        # planes that fill a fraction of the chip), the cycle pass is a few dozen chip-filling ones -- so the D-step runs on a
        # side stream UNDER the cycle pass (SMSUT_D_OVERLAP=0: one after the other).  D's weights are first needed by phase G2.
        cur = torch.cuda.current_stream()
        overlap = self._d_overlap
        side_c = self._d_side_compute and not overlap
        if overlap or side_c:
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(cur)
        d_work = None
        with torch.cuda.stream(self._side if (overlap or side_c) else cur):
            d_scal = self._run_phase("D", self._d_phase, (x_real, x_fake, modal_org, alpha), d_params)
            if self._probe:
                self._finite_probe("D", [("d_scalars", d_scal)] + [("grad " + k, p.grad) for k, p in self.D.named_parameters()])
            if not side_c:
                # D's gradient all-reduce STARTS here (asynchronous: RCCL's own stream) and is collected after the cycle pass,
                # which needs neither D's gradients nor its updated weights -- off the critical path under data parallelism
                d_work = self.d_reducer.begin()
                if d_work is None or overlap or not self._d_async:
                    self.d_reducer.finish(d_work)
                    self.d_optimizer.step()
                    d_work = None
        if self._seg_batch:
            sf = self._run_phase("G2gen", self._g2gen_phase, (x_real, vec_to, ids, y_real), list(self._alias.values()))
            st_seg, st_semi = sf[:sf.numel() // 2], sf[sf.numel() // 2:]
        else:
            st_semi = self._run_phase("G2gen", self._g2gen_phase, (x_real, vec_to, ids), list(self._alias.values()))
        if d_work is not None:
            self.d_reducer.finish(d_work)
            self.d_optimizer.step()

        d_early = [None, False]                                # [handle, started]: D's gradient all-reduce, started under G2a1 (below)

        def join_d():                                          # side_c: the D-step's graph is done -> all-reduce, Adam (main stream)
            cur.wait_stream(self._side)
            d_scal.record_stream(cur)
            self.d_reducer.finish(d_early[0] if d_early[1] else self.d_reducer.begin())
            self.d_optimizer.step()
        self.loss.reduce_stats([st_seg, st_semi] if self._semi_on else [st_seg])     # one small all-reduce (no-op at world 1)
        if side_c and self._d_early:
            # (r05) D's gradient all-reduce STARTS here -- pack on the side stream, right behind the captured D-step, the collective on
            # RCCL's own stream -- and is collected at join_d(), after G2a1: the generator's cycle-pass backward runs under it.  Issued
            # AFTER the Dice-statistics all-reduce on purpose: the communicator executes collectives in the order the host issues
            # them, and the main stream needs the statistics before G2a1 but D's gradients only after it.  Same host order on every
            # rank; the values are those of the synchronous form (SMSUT_D_EARLY_ALLREDUCE=0).
            with torch.cuda.stream(self._side):
                d_early[0], d_early[1] = self.d_reducer.begin(), True
        if self._g_split:
            # ---------------------------------------------------- G-step in three pieces (see __init__): G2a needs no D at all
            if side_c:
                ga1 = self._run_phase("G2a1", self._g2a1_phase, (st_semi, lam_t), g_params, rebind=list(self._alias.values()))
                join_d()                                      # D-step done: its all-reduce and Adam, on this stream
                self._side.wait_stream(cur)
            else:
                ga = self._run_phase("G2a", self._g2a_phase, (y_real, st_seg, st_semi, lam_t), g_params,
                                     rebind=list(self._alias.values()))
            for p in d_params:                                # D frozen: its unused gradients are neither computed nor reduced
                p.requires_grad_(False)
            with torch.cuda.stream(self._side if (overlap or side_c) else cur):   # beside generator work on the main stream
                gb = self._run_phase("G2d", self._g2d_phase, (modal_trg,), [])
            for p in d_params:
                p.requires_grad_(True)
            if side_c and self._seg_batch:
                ga2 = self._run_phase("G2a2", self._g2a2_phase, (y_real, st_seg, st_semi, lam_t), [], rebind=g_params)
            elif side_c:
                ga2 = self._run_phase("G2a2", self._g2a2_phase, (y_real, st_seg), [], rebind=g_params)
            if overlap or side_c:
                cur.wait_stream(self._side)
                d_scal.record_stream(cur)        # allocated on the side stream (eager mode), read on this one
                gb.record_stream(cur)
                if self._gx_d is not None:       # (None on a replay: nothing was allocated)
                    self._gx_d.record_stream(cur)
            if self._probe:
                self._finite_probe("D.step", list(self.D.named_parameters()))
            self._run_phase("G2c", self._g2c_phase, (), [], rebind=g_params)
            if side_c and self._seg_batch:       # [G_fake, G_rec, G_cls, G_seg, G_semi, G_nce]; ga1 = [rec, nce], ga2 = [seg, semi]
                g_scal = torch.cat([gb[0:1], ga1[0:1], gb[1:2], ga2, ga1[1:2]])
            elif side_c:
                g_scal = torch.cat([gb[0:1], ga1[0:1], gb[1:2], ga2, ga1[1:3]])
            else:
                g_scal = torch.cat([gb[0:1], ga[0:1], gb[1:2], ga[1:4]])
        else:
            if side_c:
                join_d()
            if overlap:
                cur.wait_stream(self._side)
                d_scal.record_stream(cur)        # allocated on the side stream (eager mode), read by the final cat on this one
            if self._probe:
                self._finite_probe("D.step", list(self.D.named_parameters()))

            # -------------------------------------------------------- G-step (:150-180), the part that needs the updated D
            for p in d_params:                                # D frozen: its unused gradients are neither computed nor reduced
                p.requires_grad_(False)
            g_scal = self._run_phase("G2", self._g2_phase, (y_real, modal_trg, st_seg, st_semi, lam_t), g_params,
                                     rebind=list(self._alias.values()))
            for p in d_params:
                p.requires_grad_(True)
        if self._probe:
            self._finite_probe("G2", [("g_scalars", g_scal)] + [("grad " + k, p.grad) for k, p in self.net.named_parameters()])
        ops.pair_assert_empty()                # every parked operand set of the cycle pass met its G(x_real) partner
        self.g_reducer.reduce()
        sgd_step(self.optimizer)
        if self._probe:
            self._finite_probe("G.step", list(self.net.named_parameters()))

        lr_ = self.poly_lr()                                                                   # :198-202
        for grp in list(self.optimizer.param_groups) + list(self.d_optimizer.param_groups):
            grp["lr"] = lr_
        self.iter += 1
        self._eager_done = True
        return torch.cat([d_scal, g_scal])

    def train_epoch(self, lb_loader, ul_loader, meter):
        self.net.train(); self.D.train()
        lb_itr, ul_itr = iter(lb_loader), iter(ul_loader)
        tic = time.time()
        from ..misc.utils import ScalarFetcher
        fetch = ScalarFetcher(len(SCALARS), self.device)
        try:
            for i in range(self.n_critic * cfg.num_iter_per_epoch):
                try:
                    x1, y_real, mo1, _ = next(lb_itr)
                except StopIteration:
                    lb_itr = iter(lb_loader); x1, y_real, mo1, _ = next(lb_itr)
                try:
                    x2, _, mo2, _ = next(ul_itr)
                except StopIteration:
                    ul_itr = iter(ul_loader); x2, _, mo2, _ = next(ul_itr)
                x_real = torch.cat([x1.to(self.device, non_blocking=True), x2.to(self.device, non_blocking=True)], 0)
                modal_org = torch.cat([mo1, mo2], 0)
                scal = self.train_iteration(x_real, y_real.to(self.device, non_blocking=True), modal_org)
                # the iteration's ten scalars travel to pinned host memory without a sync and are consumed ONE ITERATION LATER
                # (misc.utils.ScalarFetcher): same values in the same order for the meter and the log, no device stall
                done = fetch.push(scal, (i, self.iter, int(mo1[0])))
                if done is not None:
                    tic = self._consume_scalars(done, meter, tic)
        finally:
            # also on the way out of an exception: the last finished iteration's scalars still reach the meter / the log
            done = fetch.flush()
            if done is not None:
                self._consume_scalars(done, meter, tic)

    def _consume_scalars(self, done, meter, tic):
        vals, (i, it, modality) = done
        if meter is not None:
            v, n = meter.collect_loss_by(vals[SCALARS.index("G_seg")], modality, cfg.batch_size)
            meter.accumulate(v, n)
        if (i + 1) % (self.n_critic * self.log_step) == 0:
            # (host time between two log lines: the host runs up to one iteration ahead of the device, so this is the enqueue
            #  rate, equal to the device rate once the launch queue is full)
            self.info("Iter: %d/%d(%d), elapsed (host): %.2fs, " % (i, cfg.num_iter_per_epoch, it, time.time() - tic)
                      + " ".join("%s: %.4f," % kv for kv in zip(SCALARS, vals)))
            tic = time.time()
        return tic

    def translate_all(self, x_fixed, modal_org):
        """The per-epoch sample grid of :205-214 / saving_pseudo :216-: x translated to every modality (no grad)."""
        vec_org = self.label2onehot(modal_org, cfg.n_modal).to(self.device)
        outs = [x_fixed]
        with torch.no_grad():
            for vec in self.create_vectors(vec_org, cfg.n_modal):
                _, x_fake, _, _ = self.net(x_fixed, vec - vec_org)
                outs.append(x_fake)
        return self.denorm(torch.cat(outs, dim=3))

    def saving_pseudo(self, loader_type, expr_root):
        """:216-304 without the JPEG writing (PIL/torchvision side): returns predictions and translations per batch."""
        self.net.eval()
        _, _, loader = self.get_loaders(loader_type)
        out = []
        with torch.no_grad():
            for img, msk, mdl, inm in loader:
                img = img.to(self.device)
                grid = self.translate_all(img, mdl)
                seg, _ = self.net(img, val_phase=True)
                out.append((inm, ops.argmax_channels(seg).cpu(), grid.cpu()))
        return out


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-p", "--phase", type=str, choices=("train", "test", "pseudo"))
    p.add_argument("-f", "--fold", type=int, default=0)
    p.add_argument("-nm", "--expr_name", type=str)
    p.add_argument("-i", "--model_id", type=str, help="only for test")
    p.add_argument("-wh", "--which_ckpt", type=str, default="last")
    args = p.parse_args(argv)
    seed_all()
    trainer = UGANConsisTrainer(args.phase, args)
    if args.phase == "train":
        trainer.fit("inTurn")
    elif args.phase == "test":
        trainer.load_model(args.model_id, args.which_ckpt)
        trainer.test("inTurn", pjoin(trainer.expr_root, args.model_id))
    elif args.phase == "pseudo":
        trainer.load_model(args.model_id, args.which_ckpt)
        trainer.saving_pseudo("inTurn", pjoin(trainer.expr_root, args.model_id))


if __name__ == "__main__":
    main()
