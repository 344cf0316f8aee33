"""``BaseTrainer`` -- the caller side of the hot path (reference trainer/baseTrainer.py:33-375), kept to what
drives the kernels: device/bookkeeping, the DiceCE criterion, ``fit`` epoch loop, ``validate_epoch`` with the
last-batch padding, Dice matrix, checkpoint save/load with the reference's file names.

MI355X-first differences (all opt-in or invisible to a caller of the reference API):
  * one process per GPU: ``LOCAL_RANK`` selects the device and, when ``WORLD_SIZE > 1``, gradients are averaged
    with one flat RCCL all-reduce per network (``parallel.GradAllReducer``) instead of ``nn.DataParallel``;
  * the mains call ``fit('inTurn')`` / ``test('inTurn', ...)`` as the reference's do; ``get_loaders`` hands out the real PNG
    loaders when ``config.base_root`` holds a processed dataset and the synthetic slice source (same batch contract) otherwise;
  * TensorBoard / code snapshot / medpy are optional extras that are skipped when not installed.
"""
import abc
import logging
import os
import time
from os.path import join as pjoin

import numpy as np
import torch

from .. import config as cfg
from .. import ops, parallel
from ..misc.loss import DiceAndCrossEntropyLoss
from ..misc.synthetic import SyntheticSliceLoader
from ..misc.utils import Meter, get_mo_matrix, maybe_mkdir


def seed_all(seed=None):
    """The reference seeds random / numpy / torch with config.seed (uganConsisTrainer.py:317-320).  One process per GPU:
    rank r seeds with ``seed + r`` so the ranks draw different target modalities, interpolation weights and augmentation
    parameters (the weights themselves are broadcast from rank 0 in ``build_network``; the train sampler keeps its own
    rank-independent generator, data_loader/inTurnLoader.py)."""
    import random as _random
    s = (cfg.seed if seed is None else seed) + int(os.environ.get("RANK", "0"))
    _random.seed(s); np.random.seed(s); torch.manual_seed(s)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(s)
    return s


def make_sgd(params, lr, momentum, weight_decay):
    """torch.optim.SGD with the single-kernel ("fused") multi-tensor step: same update rule as the reference's optimizer
    (baseline trainers: SGD(lr, momentum=0.9, weight_decay)), ~3 launches instead of ~12 per step.  The parameters, their
    gradients and the momentum buffers share the HWIO strides, which is all the fused kernel needs (dense, same layout)."""
    opt = torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay, fused=True)
    opt._smsut_stepper = SgdStepper(opt)         # (class below; used by the trainers through ``sgd_step``)
    return opt


class SgdStepper:
    """``optimizer.step()`` of a ``torch.optim.SGD(momentum, weight_decay)`` as ONE launch over all parameters
    (``smsut_sgd_momentum_multi``, csrc/pointwise.hip: same update rule; torch's fused multi-tensor kernel needs four under-filled
    launches for the generator's 12.6 MB: 113 us against ~10).  The optimizer object stays the owner of the state (``momentum_buffer``
    per parameter: checkpoints and ``resume`` see what they always saw); this class only keeps a device table of {parameter, gradient,
    buffer} pointers, rebuilt when a pointer changes (under hipGraph replay and with the data-parallel bucket they never do).  Falls
    back to ``optimizer.step()`` for the first step (the buffers are created there), for an option this kernel does not implement,
    for a tensor whose layout differs from its parameter's, and -- permanently -- when the pointers keep changing (eager mode)."""

    def __init__(self, optimizer):
        self.opt = optimizer
        self._params = None          # [(parameter, momentum buffer)] of the table in use
        self._gptrs = None           # the gradient pointers that table was built for
        self._tab = None
        self._misses = 0
        self.enabled = os.environ.get("SMSUT_SGD_ONE_LAUNCH", "1") not in ("0", "")
        self.launched = 0

    def _hyper(self):
        hyper = None
        for grp in self.opt.param_groups:
            if grp.get("nesterov") or grp.get("dampening", 0) != 0 or grp.get("maximize") or grp.get("momentum", 0) == 0:
                return None
            h = (float(grp["lr"]), float(grp["momentum"]), float(grp["weight_decay"]))
            if hyper is not None and h != hyper:
                return None
            hyper = h
        return hyper

    def _build(self):
        """Validate every (parameter, gradient, buffer) triple and upload the pointer table; False: this step goes to torch."""
        from .. import _hip as H
        params, ents = [], []
        for grp in self.opt.param_groups:
            for p in grp["params"]:
                g = p.grad
                if g is None:
                    continue
                buf = self.opt.state.get(p, {}).get("momentum_buffer")
                if (buf is None or not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse
                        or g.stride() != p.stride() or buf.stride() != p.stride()):
                    return False
                params.append((p, buf))
                ents.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel()))
        if not ents:
            return False
        chunk = H.call("smsut_sgd_chunk")
        dev = params[0][0].device
        blk_ent, blk_chunk = [], []
        for i, e in enumerate(ents):
            nb = (e[3] + chunk - 1) // chunk
            blk_ent += [i] * nb
            blk_chunk += list(range(nb))
        self._tab = (torch.tensor(ents, dtype=torch.int64, device=dev), torch.tensor(blk_ent, dtype=torch.int32, device=dev),
                     torch.tensor(blk_chunk, dtype=torch.int32, device=dev), len(blk_ent))
        self._params = params
        self._gptrs = [(e[0], e[1]) for e in ents]                 # (parameter, gradient) addresses the table was built for
        return True

    def step(self):
        from .. import _hip as H
        hyper = self._hyper() if self.enabled and self._misses <= 8 else None
        if hyper is None:
            return self.opt.step()
        same = False
        if self._params is not None:
            n_with_grad = sum(1 for grp in self.opt.param_groups for p in grp["params"] if p.grad is not None)
            if n_with_grad == len(self._params):
                same = all(p.grad is not None and (p.data_ptr(), p.grad.data_ptr()) == pg
                           and self.opt.state[p].get("momentum_buffer") is buf for (p, buf), pg in zip(self._params, self._gptrs))
        if same:
            self._misses = 0
        else:
            self._misses += 1
            if not self._build():
                return self.opt.step()
        t, be, bc, nb = self._tab
        H.call("smsut_sgd_momentum_multi", t, be, bc, nb, hyper[0], hyper[1], hyper[2], torch.cuda.current_stream().cuda_stream)
        self.launched += 1


def sgd_step(optimizer):
    """``optimizer.step()`` through the optimizer's one-launch stepper (``make_sgd`` attaches it); any other optimizer: its own step."""
    st = getattr(optimizer, "_smsut_stepper", None)
    return st.step() if st is not None else optimizer.step()


def make_adam(params, lr, betas, weight_decay):
    return torch.optim.Adam(params, lr, betas, weight_decay=weight_decay, fused=True)


class BaseTrainer(abc.ABC):
    def __init__(self, phase, args=None):
        self.args = args
        self.rank, self.world, self.local_rank, self.group = parallel.init_from_env()
        if not torch.cuda.is_available():
            raise RuntimeError("SMSUT trainers run on an MI355X (torch.device('cuda')); no CPU fallback")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.phase = phase
        self.fold = 0 if args is None else getattr(args, "fold", 0)
        expr_name = getattr(args, "expr_name", None) if args is not None else None
        self.expr_root = pjoin(cfg.expr_root, expr_name or self.__class__.__name__)
        self.model_idx, self.logger, self.modality = None, None, "all"
        if self.phase == "train" and self.rank == 0 and getattr(args, "write_env", True):
            self.init_train_env(self.expr_root)
        self.net = None
        self.build_network()
        self.loss = DiceAndCrossEntropyLoss(weight_ce=cfg.weight_ce, weight_dc=cfg.weight_dc, batch_dice=True,
                                            process_group=self.group)        # baseTrainer.py:57
        self.epoch = 0
        self.iter = 0
        self._train_loaders = None                   # (lb, ul) while fit() runs: their position is part of the train state
        self._resume_loader_states = None

    # ------------------------------------------------------------------ bookkeeping (baseTrainer.py:65-123)
    @staticmethod
    def sigmoid_rampup(current, rampup_length):
        if rampup_length == 0:
            return 1.0
        current = np.clip(current, 0.0, rampup_length)
        ph = 1.0 - current / rampup_length
        return float(np.exp(-5.0 * ph * ph))

    def init_train_env(self, expr_root):
        maybe_mkdir(expr_root)
        self.model_idx = str(len(os.listdir(expr_root))).rjust(3, "0")
        root = pjoin(expr_root, self.model_idx)
        maybe_mkdir(root, *(pjoin(root, d) for d in ("ckpt", "tb", "result", "sample")))
        logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s: %(message)s")
        self.logger = logging.getLogger("FileLogger")
        self.logger.setLevel(logging.INFO)           # (basicConfig is a no-op when the host application configured logging first)
        for h in list(self.logger.handlers):         # one run directory per trainer: do not keep writing into an earlier one
            if isinstance(h, logging.FileHandler):
                self.logger.removeHandler(h); h.close()
        self.logger.addHandler(logging.FileHandler(pjoin(root, "train.log"), mode="a", encoding="utf-8"))
        self.info(f"Create train environment in {root}.")

    def info(self, s):
        if self.logger is not None:
            self.logger.info(s)
        elif self.rank == 0:
            print(s)

    @abc.abstractmethod
    def build_network(self):
        ...

    def load_model(self, model_idx=None, which_ckpt="last"):
        path = pjoin(self.expr_root, model_idx or self.model_idx, "ckpt", f"{which_ckpt}.ckpt")
        self.net.load_state_dict(torch.load(path, map_location="cpu"))
        self.info(f"Load model from {path}.")

    def save_model(self, prefix):
        """Weights by rank 0; the train state is a COLLECTIVE under data parallelism (every rank contributes its RNG and loader
        state), so all ranks call this together -- ``fit`` does."""
        if self.rank == 0:
            path = pjoin(self.expr_root, self.model_idx, "ckpt", f"{prefix}.ckpt")
            torch.save({k: v.contiguous() for k, v in self.net.state_dict().items()}, path)
            self.info(f"Save model to {path}.")
        self.save_train_state(prefix)

    # ------------------------------------------------------------------ resume (the reference saves
    # state_dicts only -- uganShp0Trainer.py:94-107, baseTrainer.py:120-123 -- and cannot continue a run)
    _OPTIMIZERS = ("optimizer", "d_optimizer", "optimizer1", "optimizer2")

    def _rank_state(self):
        """What differs from rank to rank: the host + device RNG streams (ranks seed with ``seed + rank``: target modality,
        interpolation weights, augmentation draws) and the train loaders' position (sampler generator, per-modality cursors,
        shuffled id lists)."""
        import random as _random
        loaders = [ld.state_dict() if hasattr(ld, "state_dict") else None for ld in (self._train_loaders or ())]
        return {"rank": self.rank,
                "rng": {"python": _random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state(),
                        "cuda": torch.cuda.get_rng_state(self.device)},
                "loaders": loaders}

    def save_train_state(self, prefix):
        """``{prefix}_state.ckpt`` next to the weight files: every optimizer's state (momentum / Adam moments, current LR),
        ``iter`` / ``epoch`` (poly LR, consistency ramp-up and the ``iter >= 1000`` switch depend on them) and, PER RANK, the
        host + device RNG states and the train loaders' state (gathered to rank 0: r02 saved rank 0's streams only and a resumed
        run had every rank drawing the same modalities / alphas / augmentations, and restarted the data order).  Optimizer state
        tensors keep the parameters' HWIO strides (the fused optimizers need the layouts to agree), so they are saved as they
        are, not ``.contiguous()``."""
        if self.phase != "train":
            return None
        mine = self._rank_state()
        per_rank = [mine]
        if self.world > 1:
            import torch.distributed as dist
            per_rank = [None] * self.world if self.rank == 0 else None
            dist.gather_object(mine, per_rank, dst=0, group=self.group)
        if self.rank != 0 or self.model_idx is None:
            return None
        state = {"iter": self.iter, "epoch": self.epoch, "world": self.world,
                 "optimizers": {n: getattr(self, n).state_dict() for n in self._OPTIMIZERS if hasattr(self, n)},
                 "rng": mine["rng"], "ranks": per_rank}
        path = pjoin(self.expr_root, self.model_idx, "ckpt", f"{prefix}_state.ckpt")
        torch.save(state, path)
        return path

    def resume(self, model_idx, which_ckpt="last", restore_rng=True):
        """Continue a run: weights (``load_model``), optimizer states, ``iter`` / ``epoch``, and THIS rank's RNG streams and
        loader position.  Call before the first training step (captured graphs bind gradient buffers at capture time).  When the
        world size differs from the saved one the per-rank streams cannot be mapped: the rank reseeds with
        ``seed + rank`` mixed with ``iter`` (distinct per rank, reproducible) and the loaders start a fresh order."""
        import random as _random
        self.load_model(model_idx, which_ckpt)
        self.net.to(self.device)
        path = pjoin(self.expr_root, model_idx, "ckpt", f"{which_ckpt}_state.ckpt")
        state = torch.load(path, map_location="cpu", weights_only=False)
        for n, sd in state["optimizers"].items():
            getattr(self, n).load_state_dict(sd)
        self.iter, self.epoch = int(state["iter"]), int(state["epoch"])
        ranks = state.get("ranks") or [{"rank": 0, "rng": state["rng"], "loaders": []}]
        same_world = int(state.get("world", 1)) == self.world and len(ranks) == self.world
        if restore_rng:
            if same_world:
                rng = ranks[self.rank]["rng"]
                _random.setstate(rng["python"]); np.random.set_state(rng["numpy"])
                torch.set_rng_state(rng["torch"]); torch.cuda.set_rng_state(rng["cuda"], self.device)
            else:
                seed_all(cfg.seed + 1000003 * (self.iter + 1))            # (seed_all adds the rank)
        self._resume_loader_states = ranks[self.rank]["loaders"] if same_world else None
        # (building the loaders after resume() draws from the host RNG -- the inTurn samplers shuffle with the global
        #  ``random`` at world == 1 -- so the restored streams are put back once more when the loaders are adopted)
        self._resume_rng = ranks[self.rank]["rng"] if (restore_rng and same_world) else None
        self.model_idx = self.model_idx or model_idx
        self.info(f"[*] Resumed from {path}: iter {self.iter}, epoch {self.epoch}, rank {self.rank}/{self.world}"
                  f"{'' if same_world else ' (world size changed: reseeded, fresh data order)'}.")

    def adopt_train_loaders(self, lb, ul):
        """Register the loaders whose position belongs to the train state; after ``resume()`` they continue the saved order."""
        self._train_loaders = (lb, ul)
        if self._resume_loader_states:
            for ld, st in zip(self._train_loaders, self._resume_loader_states):
                if st is not None and hasattr(ld, "load_state_dict"):
                    ld.load_state_dict(st)
            self._resume_loader_states = None
        rng = self.__dict__.get("_resume_rng")
        if rng is not None:                                  # exact continuation: the streams as they were checkpointed
            import random as _random
            _random.setstate(rng["python"]); np.random.set_state(rng["numpy"])
            torch.set_rng_state(rng["torch"]); torch.cuda.set_rng_state(rng["cuda"], self.device)
            self._resume_rng = None

    # ------------------------------------------------------------------ loaders
    def get_loaders(self, loader_type):
        """baseTrainer.py:128-135.  With ``config.base_root`` pointing at a processed PNG dataset the 'inTurn' loaders
        are the real ones (single-modality round-robin batches, joint augmentation on the device); otherwise the
        synthetic source with the same batch contract."""
        if loader_type not in ("inTurn", "base", "synthetic"):
            raise NotImplementedError
        if loader_type == "inTurn" and cfg.base_root and os.path.isdir(cfg.base_root):
            from ..data_loader import inTurnLoader as inlod
            mk = lambda phase, fold, aug: inlod.get_loader(cfg.base_root, phase, fold, cfg.batch_size, aug,
                                                           device=self.device, split_yaml=cfg.split_yaml,
                                                           rank=self.rank if phase != "test" else 0,
                                                           world=self.world if phase != "test" else 1)
            return mk("train", self.fold, cfg.data_aug), mk("val", self.fold, cfg.data_aug), mk("test", 0, None)
        if loader_type != "synthetic" and not self.__dict__.get("_told_synthetic"):
            # the reference's mains ask for 'inTurn' (uganConsisTrainer.py:320-332); with no dataset configured this is what they get
            self._told_synthetic = True
            self.info("loader %r: config.base_root %r is not a processed PNG dataset -- using the synthetic slice source "
                      "(same batch contract)" % (loader_type, cfg.base_root))
        n = getattr(self.args, "iters_per_epoch", None) or cfg.num_iter_per_epoch
        mk = lambda labeled, nb: SyntheticSliceLoader(cfg.batch_size, n_batches=nb, device=self.device,
                                                      labeled=labeled, rank=self.rank)
        return mk(True, n), mk(False, n), mk(True, max(n // 10, 1))

    # ------------------------------------------------------------------ epoch loop (baseTrainer.py:125-201)
    def fit(self, loader_type="synthetic", max_epoch=None):
        lb, ul, test = self.get_loaders(loader_type)
        self.adopt_train_loaders(lb, ul)
        keys_min = [f"loss_{i}" for i in range(cfg.n_modal)] + ["loss"]
        keys_max = [f"dice_{i}" for i in range(cfg.n_modal)] + ["dice"]
        train_meter = Meter(keys_min, [], alpha=cfg.exp_alpha)
        test_meter = Meter(keys_min, keys_max, alpha=1.0)
        tic = time.time()
        for epoch in range(self.epoch, max_epoch or cfg.max_epoch):          # (self.epoch > 0 after resume())
            train_meter.reset_cur()
            self.train_epoch(lb, ul, train_meter)
            self.epoch += 1
            train_meter.update_cur()
            self.info("[TRN] Epoch: %d/%d, elapsed: %.2fs,%s" % (epoch, cfg.max_epoch, time.time() - tic, train_meter))
            tic = time.time()
            test_meter.reset_cur()
            gt = self._collect_labels(test)
            n_prd, prd = self.validate_epoch(test, gt, test_meter)
            dices = self.validate_dice(prd, gt)
            test_meter.accumulate(dices, {k: 1.0 for k in dices})
            test_meter.update_cur()
            self.info("[TST] Epoch: %d/%d, elapsed: %.2fs,%s" % (epoch, cfg.max_epoch, time.time() - tic, test_meter))
            # save_model is collective under DP (per-rank train state): rank 0 decides, every rank follows
            best = self._agree(self.model_idx is not None and test_meter.cur_values["dice"] >= test_meter.best_values["dice"])
            if best:
                self.save_model(prefix="best")
        if self._agree(self.model_idx is not None):
            self.save_model(prefix="last")

    def _agree(self, flag):
        """Rank 0's decision, on every rank (validation data / run directories exist per rank or on rank 0 only)."""
        if self.world <= 1:
            return bool(flag)
        import torch.distributed as dist
        box = [bool(flag)]
        dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    @staticmethod
    def _collect_labels(loader):
        """Ground-truth volumes keyed 'm_pid' from a loader whose names are 'm_pid_z' (get_label_npys, utils.py:163)."""
        vols = {}
        state = (loader.gen.get_state(), loader._i) if hasattr(loader, "gen") else None
        for _, msk, _, names in loader:
            for i, nm in enumerate(names):
                m, pid, z = nm.split("_")
                vols.setdefault(f"{m}_{pid}", {})[int(z)] = msk[i].cpu().numpy()
        if state is not None:
            loader.gen.set_state(state[0]); loader._i = state[1]
        return {k: np.stack([v[z] for z in sorted(v)]) for k, v in vols.items()}

    @abc.abstractmethod
    def train_epoch(self, lb_loader, ul_loader, meter):
        ...

    def _forward_eval(self, img):
        return self.net(img)

    def validate_epoch(self, loader, npys, meter=None, save_path=None):
        """baseTrainer.py:207-244 / uganShp0Trainer.py:250-287: no-grad inference, last batch padded to
        cfg.batch_size, on-GPU argmax, volume assembly by name 'm_pid_z'."""
        self.net.eval()
        prd, n_prd = {k: np.zeros(v.shape, dtype=v.dtype) for k, v in npys.items()}, 0
        with torch.no_grad():
            for img, msk, mdl, inm in loader:
                b, c, h, w = img.shape
                img = img.to(self.device)
                if b != cfg.batch_size:
                    img = torch.cat([img, torch.zeros(cfg.batch_size - b, c, h, w, device=self.device)], 0)
                out = self._forward_eval(img)
                if b != cfg.batch_size:
                    out = out[:b]
                loss = self.loss(out, msk.to(self.device))
                if meter is not None:
                    v, n = meter.collect_loss_by(loss.item(), mdl[0].item(), img.size(0))
                    meter.accumulate(v, n)
                pred = ops.argmax_channels(out).cpu().numpy()
                for i in range(b):
                    m, pid, z = inm[i].split("_")
                    prd[f"{m}_{pid}"][int(z)] = pred[i]
                    n_prd += 1
        return n_prd, prd

    def validate_dice(self, prd_npys, gt_npys):
        mo = get_mo_matrix(prd_npys, gt_npys)
        d = {f"dice_{i}": mo[i, -1] for i in range(cfg.n_modal)}
        d["dice"] = mo[-1, -1]
        return d

    def test(self, loader_type, expr_root):
        """baseTrainer.py:254-318 minus ASSD / connected components (third-party CPU post-processing, out of scope)."""
        _, _, loader = self.get_loaders(loader_type)
        gt = self._collect_labels(loader)
        n, prd = self.validate_epoch(loader, gt)
        mo = get_mo_matrix(prd, gt)
        maybe_mkdir(expr_root)
        np.savetxt(pjoin(expr_root, "dice_matrix.csv"), mo, delimiter=",", fmt="%.6f")
        return mo

    def close(self):
        """Drop every captured phase of this trainer NOW, outside any capture (``graphs.GraphedPhase.close``): graph execs are
        freed at a chosen moment, never by the cyclic collector in the middle of somebody else's capture.
        The trainer stays usable: the next step captures again.  Returns the number of phases that were open."""
        from .. import graphs
        n = 0
        g = self.__dict__.get("_graphs")
        if isinstance(g, dict):
            n += graphs.close_all(g)
            g.clear()
        g = self.__dict__.get("_graph")
        if g is not None:
            n += graphs.close_all([g])
            self._graph = None
        for k in ("_g1", "_g2", "_gx_d"):
            if k in self.__dict__:
                self.__dict__[k] = None
        return n

    def poly_lr(self):
        return cfg.lr * (1.0 - self.iter / (cfg.max_epoch * cfg.num_iter_per_epoch)) ** 0.9
