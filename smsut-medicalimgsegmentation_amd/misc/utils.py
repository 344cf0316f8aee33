"""The pieces of the reference's ``misc/utils.py`` that the hot-path trainers touch: ``Meter`` (:58-160) and
``maybe_mkdir``; plus the Dice matrix of ``get_mo_matrix`` (:180-203) with medpy's ``dc`` restated from its
formula (medpy is a third-party dependency that is not installed here; SURVEY.md 8c: parity unpinned)."""
import os
from collections import OrderedDict
from copy import deepcopy

import numpy as np

from .. import config as cfg


def maybe_mkdir(*paths):
    for p in paths:
        os.makedirs(p, exist_ok=True)


def binary_dc(result, reference):
    """medpy.metric.binary.dc: 2|A&B| / (|A|+|B|); 0.0 when both are empty."""
    a = np.asarray(result).astype(bool)
    b = np.asarray(reference).astype(bool)
    den = np.count_nonzero(a) + np.count_nonzero(b)
    return 2.0 * np.count_nonzero(a & b) / float(den) if den else 0.0


def get_mo_matrix(prd_npys, gt_npys):
    """Modality x organ Dice matrix, per 3-D volume then averaged (utils.py:180-203)."""
    matrix = np.zeros((cfg.n_modal, cfg.n_label))
    n = np.zeros((cfg.n_modal, 1))
    for k in gt_npys.keys():
        m = cfg.Modality[k.split("_")[0]].value
        p, g = prd_npys[k], gt_npys[k]
        for i in range(cfg.n_label):
            matrix[m][i] += binary_dc(p == i + 1, g == i + 1)
        n[m] += 1
    n[n == 0] += 1e-8
    matrix /= n
    full = np.zeros((cfg.n_modal + 1, cfg.n_label + 1))
    full[:cfg.n_modal, :cfg.n_label] = matrix
    full[-1, :] = np.mean(full[0:cfg.n_modal], axis=0)
    full[:, -1] = np.mean(full[:, 0:cfg.n_label], axis=1)
    return full


class Meter:
    """Running per-modality loss / dice with best tracking (utils.py:58-160), trimmed to what the trainers call."""

    def __init__(self, min_better_keys, max_better_keys, alpha=1.0):
        self.configs = OrderedDict([(k, "min") for k in min_better_keys] + [(k, "max") for k in max_better_keys])
        self.alpha = alpha
        self.best_values = self._zeros()
        self.pre_values = None
        self.cur_values = self._zeros()
        self.n = self._zeros()

    def _zeros(self):
        return {k: 0 for k in self.configs}

    @staticmethod
    def collect_loss_by(sample_loss, modal_id, n):
        k = "loss_" + str(modal_id)
        return {"loss": sample_loss * n, k: sample_loss * n}, {"loss": n, k: n}

    def __repr__(self):
        out = ""
        for k in self.configs:
            name = f"{k.split('_')[0]}_{cfg.Modality(int(k.split('_')[1])).name}" if "_" in k else k
            out += " %s: %.4f/%.4f," % (name, self.cur_values[k], self.best_values[k])
        return out

    def accumulate(self, values, n):
        for k, v in values.items():
            self.cur_values[k] += v
            self.n[k] += n[k]

    def update_cur(self, reset_best=False):
        for k in self.configs:
            if self.n[k] != 0:
                self.cur_values[k] /= self.n[k]
            if self.pre_values is not None:
                self.cur_values[k] = (1.0 - self.alpha) * self.pre_values[k] + self.alpha * self.cur_values[k]
        if self.pre_values is None or reset_best:
            self.best_values = deepcopy(self.cur_values)
            self.pre_values = deepcopy(self.cur_values)
        else:
            for k, f in self.configs.items():
                better = self.cur_values[k] < self.best_values[k] if f == "min" else self.cur_values[k] > self.best_values[k]
                if better:
                    self.best_values[k] = self.cur_values[k]
                self.pre_values[k] = self.cur_values[k]

    def reset_cur(self):
        self.cur_values = self._zeros()
        self.n = self._zeros()


class ScalarFetcher:
    """Per-iteration logging without stalling the device: the reference reads its loss scalars with ``.item()`` right after every
    iteration (trainer/uganConsisTrainer.py:148-149,157,183-188 -- 11 host syncs), which keeps the host from enqueueing iteration
    i + 1 while i runs.  Here the scalars of iteration i go to pinned host memory with a non-blocking copy and are HANDED OUT ONE
    ITERATION LATER (``push`` returns the previous item, whose copy has long finished); ``flush`` returns the last one.  Same values,
    same order -- the meter / log just see them one iteration late, and the device never waits for the host."""

    def __init__(self, numel: int, device, check_finite: bool = True):
        """``check_finite``: a NaN / Inf among the fetched scalars raises ``FloatingPointError`` naming the item's tag -- one
        iteration after it happened, but it happens: the kernels' own maxima (fmaxf) pass over NaNs silently."""
        import torch
        self._torch = torch
        self._check = check_finite
        self._bufs = [torch.empty(numel, dtype=torch.float32).pin_memory() if torch.device(device).type == "cuda"
                      else torch.empty(numel, dtype=torch.float32) for _ in range(2)]
        self._events = [None, None]
        self._tags = [None, None]
        self._cur = 0
        self._pending = None

    def _take(self, slot):
        if self._events[slot] is not None:
            self._events[slot].synchronize()
        vals = self._bufs[slot].tolist()
        if self._check and not all(v == v and abs(v) != float("inf") for v in vals):
            raise FloatingPointError(f"non-finite training scalars {vals} at {self._tags[slot]!r}")
        return vals, self._tags[slot]

    def push(self, scalars, tag=None):
        """Enqueue the device -> host copy of ``scalars`` (1-D float tensor); returns ``(values, tag)`` of the PREVIOUS push or None."""
        torch = self._torch
        prev = self._take(self._pending) if self._pending is not None else None
        slot = self._cur
        self._bufs[slot].copy_(scalars.detach().reshape(-1).float(), non_blocking=True)
        if scalars.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(scalars.device))
            self._events[slot] = ev
        self._tags[slot] = tag
        self._pending, self._cur = slot, 1 - slot
        return prev

    def flush(self):
        if self._pending is None:
            return None
        out = self._take(self._pending)
        self._pending = None
        return out
