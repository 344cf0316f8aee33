"""hipGraph capture of a trainer phase (forward + backward of one loss), via ``torch.cuda.CUDAGraph``.

The hot loop launches ~3000 small kernels per uganConsis iteration; at 8+8 slices per GPU the host cannot keep an
MI355X fed launch by launch (17 % of the step was GPU-idle launch gaps in the r01 trace).  A phase's launches are
recorded once into a hipGraph and replayed: every kernel of ``ops`` is enqueued on the current (capturing) stream,
workspaces come from the graph's private pool, and nothing in a phase synchronises with the host.

What stays OUTSIDE the graphs, by design: RNG draws (alpha ~ randn, patch ids ~ randperm), the gradient all-reduce,
the optimizer steps and the poly-LR write (Python floats), so their semantics are exactly the eager ones.
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, Sequence

import torch
import torch.distributed


def graphs_enabled(world: int, collective_free: bool = False) -> bool:
    """Default: every phase is captured, on one GPU and under data parallelism alike -- the trainers split their steps
    at the collectives (gradient all-reduce, Dice-statistics all-reduce), which run eagerly BETWEEN replays, so no
    RCCL call ever sits inside a captured region.  ``SMSUT_GRAPH=0`` is the explicit eager mode (``world`` and
    ``collective_free`` are kept for callers that want a per-phase policy)."""
    v = os.environ.get("SMSUT_GRAPH")
    if v is not None:
        return v not in ("0", "", "false", "False")
    return True


class GraphedPhase:
    """Captures ``fn(*tensors) -> tensor | tuple of tensors`` and replays it on new inputs.  A phase may end in
    ``.backward()`` (its parameters' ``.grad`` then live in the graph's pool and are rewritten by every replay) or leave
    an autograd graph behind for a LATER phase to differentiate through (the later phase's capture walks it once; on
    replay only the recorded kernels run, in capture order, on the same addresses)."""

    def __init__(self, fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor],
                 grad_params: Iterable[torch.nn.Parameter], warmup: int = 2):
        self.fn = fn
        self.params = list(grad_params)
        self.static_in = [t.detach().clone() for t in example_inputs]
        if warmup:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._clear()
                    fn(*self.static_in)
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._clear()
        self.graph = torch.cuda.CUDAGraph()
        # under torch.distributed the RCCL watchdog thread polls events while we capture: with the default "global" error
        # mode any such call from ANOTHER thread can invalidate the capture; "thread_local" polices this thread only
        mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.static_out = fn(*self.static_in)
        self.graph.replay()          # capture records without executing: run it once so static_out holds real values

    def _clear(self):
        for p in self.params:
            p.grad = None

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for s, i in zip(self.static_in, inputs):
            s.copy_(i, non_blocking=True)
        self.graph.replay()
        return self.static_out
