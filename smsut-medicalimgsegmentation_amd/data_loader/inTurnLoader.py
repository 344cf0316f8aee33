"""Single-modality round-robin batch samplers and the PNG slice dataset of the reference's input pipeline
(data_loader/inTurnLoader.py:15-97, data_loader/balanceLoader.py:17-69, baseLoader.py:87-112), with the joint
geometric augmentation moved from PIL worker processes to ONE device kernel per batch (``gpu_augment.GpuJointAugment``).

Batch contract (what every trainer consumes): ``(img fp32 [B,1,H,W] in [-1,1], msk int64 [B,H,W], modality int64 [B],
names list[str] 'm_pid_z')`` -- every batch holds slices of ONE modality, modalities taking turns.

The samplers draw from Python's ``random`` in exactly the order the reference does (one ``shuffle`` per modality at
construction, one per wrap-around), so a seeded run visits the same slice ids.
"""
import os
import random
from os.path import join as pjoin
from typing import List

import numpy as np
import torch

from .. import config as cfg
from .gpu_augment import GpuJointAugment


class InTurnTrainBatchSampler:
    """inTurnLoader.py:15-59.

    Data parallelism (one process per GPU; the reference has none): every rank builds the SAME sampler over a global batch
    of ``world * batch_size`` slices of one modality -- drawn from a private generator seeded identically on all ranks --
    and keeps its own ``batch_size``-slice share, so the ranks train on disjoint slices while the modality turn-taking
    stays aligned (r01 had every rank draw identical batches).  With ``world == 1`` the draws come from Python's global
    ``random`` in the reference's order."""

    def __init__(self, samples: List[List[int]], batch_size: int, shuffle: bool, rank: int = 0, world: int = 1, seed=None):
        self.samples = [list(s) for s in samples]
        self.num_modality = len(samples)
        self.batch_size = batch_size
        self.rank, self.world = rank, world
        self.global_batch = batch_size * world
        self.rng = random if world == 1 else random.Random(cfg.seed if seed is None else seed)
        self.starts = [0 for _ in range(self.num_modality)]
        self.shuffle = shuffle
        self.queue = list(range(self.num_modality))
        self.cur_modality = 0
        max_batch_per_modality = 0
        gb = self.global_batch
        for i, spl in enumerate(self.samples):
            n = len(spl) // gb - 1 if len(spl) % gb else len(spl) // gb                          # :31
            max_batch_per_modality = max(n, max_batch_per_modality)
            self.rng.shuffle(self.samples[i])
        self.n = self.num_modality * max_batch_per_modality

    def __iter__(self):
        gb = self.global_batch
        for _ in range(self.n):
            cur = self.queue[self.cur_modality] if self.shuffle else self.cur_modality
            s = self.starts[cur]
            if s + gb >= len(self.samples[cur]):                         # wrap: restart this modality, reshuffled (:44-47)
                self.starts[cur] = 0
                s = 0
                self.rng.shuffle(self.samples[cur])
            else:
                self.starts[cur] += gb
            batch = self.samples[cur][s: s + gb]
            if len(batch) == gb:
                yield batch[self.rank * self.batch_size: (self.rank + 1) * self.batch_size]
            if self.shuffle and self.cur_modality + 1 == self.num_modality:
                self.rng.shuffle(self.queue)
            self.cur_modality = (self.cur_modality + 1) % self.num_modality

    def __len__(self):
        return self.n

    def state_dict(self):
        """Position in the data order (part of the resumable train state): per-modality cursors and shuffled id lists, the turn
        queue, and the private generator (``world > 1``; at ``world == 1`` the draws come from Python's global ``random``, which
        the trainer saves itself)."""
        return {"samples": [list(s) for s in self.samples], "starts": list(self.starts), "queue": list(self.queue),
                "cur_modality": self.cur_modality, "rng": None if self.rng is random else self.rng.getstate()}

    def load_state_dict(self, st):
        assert [len(s) for s in st["samples"]] == [len(s) for s in self.samples], "sampler state belongs to another dataset split"
        self.samples = [list(s) for s in st["samples"]]
        self.starts, self.queue, self.cur_modality = list(st["starts"]), list(st["queue"]), int(st["cur_modality"])
        if st["rng"] is not None and self.rng is not random:
            self.rng.setstate(st["rng"])


class InTurnTestBatchSampler:
    """inTurnLoader.py:62-79: every slice once, modality after modality, last batch of a modality may be short."""

    def __init__(self, samples: List[List[int]], batch_size: int):
        self.samples, self.num_modality, self.batch_size = samples, len(samples), batch_size
        self.n = sum(len(spl) // batch_size for spl in samples)

    def __iter__(self):
        for spl in self.samples:
            for i in range(0, len(spl), self.batch_size):
                yield spl[i: i + self.batch_size]

    def __len__(self):
        return self.n


class BalanceDataset:
    """balanceLoader.py:17-69 with ``load_in_ram=True`` semantics: the slices listed in ``<root>/<split_yaml>`` for
    (phase, fold) are read once into two uint8 tensors; ``modal_sample_ids[m]`` lists the sample ids of modality m."""

    def __init__(self, data_root, phase, fold=0, split_yaml="semi-1910.yaml"):
        import yaml
        from PIL import Image
        self.data_root, self.phase, self.fold = data_root, phase, fold
        self.modal = list(cfg.Modality.__members__)
        with open(pjoin(data_root, split_yaml)) as f:
            split = yaml.load(f, Loader=yaml.FullLoader)
        imgs, msks, self.modality, self.names = [], [], [], []
        self.modal_sample_ids = [[] for _ in self.modal]
        n = 0
        for m in self.modal:
            pids = split[m][phase] if phase == "test" else split[m][phase][fold]
            for pid in pids:
                root = pjoin(data_root, m, pid, "images")
                for png in sorted(os.listdir(root)):
                    ip = pjoin(root, png)
                    imgs.append(np.array(Image.open(ip)))
                    msks.append(np.array(Image.open(ip.replace("images", "labels"))))
                    self.modality.append(cfg.Modality[m].value)
                    self.names.append(png.replace(".png", ""))
                    self.modal_sample_ids[cfg.Modality[m].value].append(n)
                    n += 1
        self.n = n
        self.images = torch.from_numpy(np.stack(imgs)) if n else torch.zeros(0, 1, 1, dtype=torch.uint8)
        self.labels = torch.from_numpy(np.stack(msks)) if n else torch.zeros(0, 1, 1, dtype=torch.uint8)

    def __len__(self):
        return self.n


class InTurnLoader:
    """Iterable over batches: gathers the sampler's ids from the in-RAM uint8 arrays, uploads them (pinned, async),
    converts to the ToTensor + Normalize(0.5, 0.5) range on the device (baseLoader.py:89-90) and applies the joint
    augmentation there."""

    def __init__(self, dataset, batch_sampler, device, augment: GpuJointAugment = None):
        self.ds, self.sampler, self.device, self.augment = dataset, batch_sampler, torch.device(device), augment

    def __len__(self):
        return len(self.sampler)

    def state_dict(self):
        return self.sampler.state_dict() if hasattr(self.sampler, "state_dict") else None

    def load_state_dict(self, st):
        if st is not None and hasattr(self.sampler, "load_state_dict"):
            self.sampler.load_state_dict(st)

    def __iter__(self):
        for ids in self.sampler:
            idx = torch.as_tensor(ids, dtype=torch.int64)
            img8, msk8 = self.ds.images[idx], self.ds.labels[idx]
            if self.device.type == "cuda":
                img8 = img8.pin_memory().to(self.device, non_blocking=True)
                msk8 = msk8.pin_memory().to(self.device, non_blocking=True)
            img = img8.float().div_(255.0).unsqueeze(1)                                   # ToTensor's [0, 1] scale
            msk = msk8.to(torch.int64)
            if self.augment is not None:                                                  # joint augmentation BEFORE Normalize, as
                img, msk = self.augment(img, msk)                                         # baseLoader.py:92-108 (fill = 0 = black)
            img = img.sub_(0.5).div_(0.5)                                                 # Normalize(0.5, 0.5)
            mdl = torch.tensor([self.ds.modality[i] for i in ids], dtype=torch.int64)
            yield img, msk, mdl, [self.ds.names[i] for i in ids]


def get_loader(data_root, phase, fold, batch_size, data_aug=None, load_in_ram: bool = True, device="cuda",
               split_yaml="semi-1910.yaml", rank: int = 0, world: int = 1):
    """inTurnLoader.py:82-97 (``load_in_ram`` is always on here; ``data_aug`` is the dict of config.py:60-71)."""
    ds = BalanceDataset(data_root, phase, fold, split_yaml)
    if phase in ("train", "val"):
        sampler = InTurnTrainBatchSampler(ds.modal_sample_ids, batch_size, shuffle=False, rank=rank, world=world)
        aug = GpuJointAugment(data_aug, cfg.input_size) if data_aug else None
    else:
        sampler = InTurnTestBatchSampler(ds.modal_sample_ids, batch_size)
        aug = None
    return InTurnLoader(ds, sampler, device, aug)
