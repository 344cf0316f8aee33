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
import weakref
from typing import Callable, Iterable, Sequence

import torch
import torch.distributed


_CAPTURING = 0           # > 0 while a GraphedPhase capture is recording on this thread
CAPTURE_SEQ = 0          # number of GraphedPhase captures started so far (ops keys per-capture scratch on it)
_LIVE = []               # weakrefs to the captured phases (to mark them for re-binding, see GraphedPhase)


def graphs_enabled() -> bool:
    """Default: every phase is captured, on one GPU and under data parallelism alike -- the trainers split their steps
    at the collectives (gradient all-reduce, Dice-statistics all-reduce), which run eagerly BETWEEN replays, so no
    RCCL call ever sits inside a captured region (enforced: ``assert_no_capture`` in every collective call site).
    ``SMSUT_GRAPH=0`` is the explicit eager mode."""
    v = os.environ.get("SMSUT_GRAPH")
    if v is not None:
        return v not in ("0", "", "false", "False")
    return True


def capturing() -> bool:
    """True while launches are being recorded into a hipGraph instead of executed (a GraphedPhase capture, or anyone else's
    capture on the current stream)."""
    return _CAPTURING > 0 or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


def assert_no_capture(what: str):
    """Called by every collective of the package (gradient all-reduce, Dice-statistics all-reduce, parameter broadcast):
    a collective issued while a phase is being captured would be baked into a hipGraph (or silently dropped on replay) --
    a trainer that wraps such a step in ``GraphedPhase`` fails here, loudly, at its first capture."""
    if _CAPTURING:
        raise RuntimeError(f"{what} was issued inside a hipGraph capture: phases must be split at their collectives "
                           f"(see trainer/uganConsisTrainer.py) or run with SMSUT_GRAPH=0")


def invalidate_grad_bindings():
    """An eager step (``p.grad = None`` + fresh gradient tensors) or anything else that re-points ``.grad`` outside a
    capture calls this: the next replay of every captured phase first puts ITS gradient buffers back."""
    for r in list(_LIVE):
        g = r()
        if g is None:
            _LIVE.remove(r)
        else:
            g._dirty = True


class GraphedPhase:
    """Captures ``fn(*tensors) -> tensor | tuple of tensors`` and replays it on new inputs.  A phase may end in
    ``.backward()`` (its parameters' ``.grad`` then live in the graph's pool and are rewritten by every replay) or leave
    an autograd graph behind for a LATER phase to differentiate through (the later phase's capture walks it once; on
    replay only the recorded kernels run, in capture order, on the same addresses).

    ``grad_params`` are cleared before the capture; they and ``rebind_params`` (parameters whose ``.grad`` this phase's
    backward fills without owning the clearing, e.g. the generator's aliases in phase G2) have the gradient tensors the
    capture bound to them RECORDED, and a replay re-installs those whenever another capture or an eager step has re-pointed
    ``.grad`` since (replaying an older graph wrote gradients to addresses ``p.grad`` no longer named, and the
    all-reduce / optimizer silently consumed stale ones)."""

    def __init__(self, fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor],
                 grad_params: Iterable[torch.nn.Parameter], warmup: int = 2,
                 rebind_params: Iterable[torch.nn.Parameter] = ()):
        global _CAPTURING, CAPTURE_SEQ
        # ``fn`` (normally a bound method of the trainer that owns this object) is used for the warm-up and the capture only and
        # is NOT kept: trainer -> GraphedPhase -> fn -> trainer was a reference cycle, so a trainer's graph execs were freed only
        # by the cyclic collector, at a moment nobody chose (r03: an abort when that moment fell into another capture).  Without
        # the cycle, ``del trainer`` (or ``trainer.close()``) frees them right there.
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
        # The cyclic garbage collector must not run while a capture records: it may finalise someone else's graphs / streams /
        # events, and destroying a graph exec while another stream captures aborts the process (seen once in r03's full test
        # suite, inside this constructor; this package's own phases no longer depend on the collector -- see above -- but user
        # objects may).  torch.cuda.graph() collects once on entry; from there to the end of the capture the collector stays off.
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        _CAPTURING += 1
        CAPTURE_SEQ += 1
        try:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.static_out = fn(*self.static_in)
        finally:
            _CAPTURING -= 1
            if gc_was_on:
                gc.enable()
        seen, self._bound, self._gmap = set(), [], {}
        for p in list(self.params) + list(rebind_params):
            if id(p) not in seen and p.grad is not None:
                seen.add(id(p))
                self._bound.append((p, p.grad))
                self._gmap[id(p)] = p.grad
        # this capture re-pointed the .grad of its parameters: older graphs that bound a DIFFERENT tensor to one of them
        # re-install their own buffers before their next replay (and mark this one in turn)
        self._dirty = False
        self._ids = seen
        for r in list(_LIVE):
            g = r()
            if g is None:
                _LIVE.remove(r)
            elif self._conflicts(g):
                g._dirty = True
        _LIVE.append(weakref.ref(self))
        self.graph.replay()          # capture records without executing: run it once so static_out holds real values

    def _clear(self):
        for p in self.params:
            p.grad = None

    def _conflicts(self, other) -> bool:
        """True when installing THIS phase's gradient tensors re-points a parameter ``other`` writes through another tensor.
        Phases that recorded the SAME tensor for every shared parameter (G2a1 / G2a2 / G2c accumulate into one buffer) leave each
        other alone -- ADVICE r03: marking on shared ids alone never settled, every replay re-assigned every .grad."""
        return any(other._gmap[i] is not self._gmap[i] for i in (self._ids & other._ids))

    def close(self):
        """Free the graph exec and its pool NOW (idempotent).  Never inside a capture: destroying an exec while a stream records
        aborts the process."""
        if getattr(self, "graph", None) is None:
            return
        if capturing():
            raise RuntimeError("GraphedPhase.close() inside a hipGraph capture")
        torch.cuda.synchronize()
        self.graph.reset()
        self.graph = None
        self.static_in = self.static_out = None
        self._bound, self._gmap, self.params, self._ids = [], {}, [], set()
        for r in list(_LIVE):
            if r() is self or r() is None:
                _LIVE.remove(r)

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        if self._dirty:
            for p, g in self._bound:
                p.grad = g
            self._dirty = False
            for r in _LIVE:                              # ... which un-binds every graph that bound OTHER tensors to them
                o = r()
                if o is not None and o is not self and self._conflicts(o):
                    o._dirty = True
        for s, i in zip(self.static_in, inputs):
            s.copy_(i, non_blocking=True)
        self.graph.replay()
        return self.static_out


def close_all(phases) -> int:
    """``close()`` every GraphedPhase in ``phases`` (a dict's values, a list, nested tuples) and return how many were open -- what
    ``BaseTrainer.close()`` calls; outside any capture."""
    n = 0
    stack = list(phases.values()) if isinstance(phases, dict) else list(phases)
    while stack:
        o = stack.pop()
        if isinstance(o, GraphedPhase):
            n += o.graph is not None
            o.close()
        elif isinstance(o, (tuple, list)):
            stack.extend(o)
    return n
