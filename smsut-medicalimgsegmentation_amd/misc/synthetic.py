"""Synthetic slice source honouring the reference's loader batch contract
(data_loader/balanceLoader.py:59-69, baseLoader.py:89-90): ``(img fp32 [B,1,H,W] in [-1,1], msk int64 [B,H,W],
modality int64 [B], names list[str] 'm_pid_z')`` with ONE modality per batch, cycling round-robin
(data_loader/inTurnLoader.py:37-57).  The dataset pipeline itself is out of the hot-path scope;
benchmarks and tests use this generator, seeded from ``config.seed``."""
import torch

from .. import config as cfg


class SyntheticSliceLoader:
    def __init__(self, batch_size, size=None, n_classes=None, n_batches=None, device="cuda", seed=None, labeled=True,
                 block=16, rank=0):
        self.bs, self.size = batch_size, size or cfg.input_size
        self.ncls = n_classes or (cfg.n_label + 1)
        self.n_batches = n_batches or cfg.num_iter_per_epoch
        self.device, self.labeled, self.block = torch.device(device), labeled, block
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed((seed if seed is not None else cfg.seed) + 7919 * rank + (0 if labeled else 1))
        self._i = 0

    def __len__(self):
        return self.n_batches

    def state_dict(self):
        return {"gen": self.gen.get_state(), "i": self._i}

    def load_state_dict(self, st):
        self.gen.set_state(st["gen"]); self._i = int(st["i"])

    def _batch(self):
        b, s = self.bs, self.size
        img = (0.5 * torch.randn(b, 1, s, s, generator=self.gen)).clamp_(-1, 1)
        blk = max(s // self.block, 1)
        small = torch.randint(0, self.ncls, (b, blk, blk), generator=self.gen)
        msk = small.repeat_interleave(s // blk, 1).repeat_interleave(s // blk, 2)
        m = self._i % cfg.n_modal
        mod = torch.full((b,), m, dtype=torch.int64)
        names = [f"{cfg.Modality(m).name}_{self._i:03d}_{z}" for z in range(b)]
        self._i += 1
        return img.to(self.device, non_blocking=True), msk.to(self.device, non_blocking=True), mod, names

    def __iter__(self):
        for _ in range(max(self.n_batches, 1)):
            yield self._batch()
