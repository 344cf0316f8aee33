"""Data parallelism for the SMSUT step: one process per GPU, RCCL (``backend='nccl'`` on ROCm) over xGMI.

The path shards as pure data parallel: every slice is independent in all convs and in
InstanceNorm, so the only exchange is ONE all-reduce of the flattened fp32 gradients per network per
optimizer step (D after ``d_loss.backward()``: 9.68 MB; G after ``g_loss.backward()``: 12.59 MB; U-Net
8.13 MB) plus the optional 3xC-float Dice statistics all-reduce in ``ops.DiceCEFn``.  Payloads this small
are latency-bound on xGMI (7 links x ~153 GB/s), so one flat bucket per network beats per-tensor calls.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def force_dist() -> bool:
    """``SMSUT_FORCE_DIST=1``: run the collective code path even at world size 1 (a one-rank RCCL communicator) -- the
    rehearsal of the multi-GPU path that a one-GPU box allows: communicator set-up, flat-buffer all-reduces on the side
    stream, hipGraph capture next to the RCCL watchdog thread.  Numerically the identity."""
    return os.environ.get("SMSUT_FORCE_DIST", "0") not in ("0", "")


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank, group or None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU boxes): SMSUT_DIST_BACKEND=gloo runs the N>1 path over gloo with device tensors,
    # SMSUT_FORCE_DEVICE=0 puts every rank on the same card
    forced = os.environ.get("SMSUT_FORCE_DEVICE")
    if forced is not None:
        local = int(forced)
    if world <= 1 and not force_dist():
        return 0, 1, local, None
    world = max(world, 1)
    if backend is None:
        backend = os.environ.get("SMSUT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local, dist.group.WORLD


def _flat_memory(t: torch.Tensor) -> torch.Tensor:
    """1-D view over a dense (possibly permuted) tensor's memory, in memory order."""
    return t.as_strided((t.numel(),), (1,), t.storage_offset())


class GradAllReducer:
    """Averages the gradients of ``params`` across ranks with one flat all-reduce.

    ``reduce()`` is called after ``backward()`` and before ``optimizer.step()``.  Parameters whose ``.grad``
    is None on this rank contribute zeros (keeps the collective shape identical on every rank)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1
        self.numel = sum(p.numel() for p in self.params)
        self._flat: Optional[torch.Tensor] = None

    def reduce(self):
        """Synchronous form: pack, all-reduce, average, unpack."""
        self.finish(self.begin())

    def begin(self):
        """Pack the gradients and START the all-reduce (``async_op=True``: RCCL runs it on its own stream behind the work
        already queued on the current one).  Whatever the caller launches next overlaps with it; ``finish`` waits, averages and
        unpacks.  Returns a handle (None when there is nothing to reduce)."""
        if self.world <= 1 and not (force_dist() and self.group is not None):
            return None
        from . import graphs
        graphs.assert_no_capture("GradAllReducer.begin (gradient all-reduce)")
        dev = self.params[0].device
        if self._flat is None or self._flat.device != dev:
            self._flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        flat = self._flat
        if all(p.grad is not None for p in self.params):
            # one batched concat (two launches for 175 tensors): 26 us for the generator's 12.6 MB, 75 us as a multi-tensor copy
            torch.cat([_flat_memory(p.grad) for p in self.params], out=flat)
        else:
            off = 0
            views, srcs = [], []
            for p in self.params:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    views.append(flat[off:off + n])
                    srcs.append(_flat_memory(p.grad))
                off += n
            if views:
                torch._foreach_copy_(views, srcs)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, work):
        if work is None:
            return
        work.wait()                                  # the current stream now waits for the collective
        flat = self._flat
        flat.mul_(1.0 / self.world)
        off = 0
        dsts, chunks = [], []
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)       # preserves the parameter's (permuted) strides
            dsts.append(_flat_memory(p.grad))
            chunks.append(flat[off:off + n])
            off += n
        torch._foreach_copy_(dsts, chunks)


def broadcast_parameters(module: torch.nn.Module, group=None, src: int = 0):
    """Make every rank start from rank ``src``'s weights."""
    if group is None or not dist.is_initialized() or (dist.get_world_size(group) <= 1 and not force_dist()):
        return
    from . import graphs
    graphs.assert_no_capture("broadcast_parameters")
    for t in list(module.parameters()) + list(module.buffers()):
        flat = _flat_memory(t.data)
        dist.broadcast(flat, src=src, group=group)
