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


def graphs_enabled(world: int, collective_free: bool = False) -> bool:
    """Default: every phase on single-GPU runs; under data parallelism only the phases that contain no collective
    (``collective_free``: the D-step of the GAN trainers -- its gradient all-reduce runs after the phase, outside the
    graph).  A phase with the Dice-statistics all-reduce inside (G-step, U-Net step) stays
    eager under DP: RCCL inside a captured hipGraph has not been exercised on hardware.  ``SMSUT_GRAPH=0/1`` overrides
    (1 = every phase, whatever the world size; ``dp`` = the data-parallel policy on any world size)."""
    v = os.environ.get("SMSUT_GRAPH")
    if v == "dp":                      # the data-parallel policy on any world size (to measure it on one GPU)
        return collective_free
    if v is not None:
        return v not in ("0", "", "false", "False")
    return world == 1 or collective_free


class GraphedPhase:
    """Captures ``fn(*tensors) -> tensor`` (which ends in ``.backward()``) and replays it on new inputs."""

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
        with torch.cuda.graph(self.graph):
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
