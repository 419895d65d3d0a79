"""hipGraph capture of a whole optimisation step.

The reference removes its per-op launch overhead with ``torch.compile(mode='reduce-overhead')`` (CUDA graphs under a
tracing compiler; maxvit_rnn.py:43-51).  Here nothing is traced: every kernel behind include/rvt_hip.h takes an explicit
stream, allocates nothing and synchronises nothing, so the step — prepack, weight pack, four stage-major forwards, the
BPTT backward with its weight-gradient side stream, the gradient fold, (the RCCL all-reduces,) the fused optimizer — is
captured once as it is issued and replayed as ONE graph launch.  Shapes are static per (T, B, resolution) bucket, which
is how the reference trains (fixed sequence length and batch size, config/experiment/*).

    step = GraphedStep(lambda: train_step(static_batch))     # warm-up runs + capture
    static_batch.copy_(next_batch); step()                   # replay
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedStep:
    def __init__(self, fn: Callable[[], None], warmup: int = 3, device: Optional[torch.device] = None, models=()):
        """fn must read its inputs from tensors that stay alive and keep their addresses (copy new data INTO them), must
        not synchronise with the host, and should leave parameter gradients in place (rvt_amd assigns persistent bucket
        views to ``.grad``; use ``optimizer.zero_grad(set_to_none=True)`` inside fn)."""
        self.fn = fn
        # models whose kernel-side weight copies the captured step re-packs: a replay updates the parameters without bumping
        # their version counters on the host, so the next inference forward must not trust its cache
        self.models = tuple(models)
        # SIDE EFFECT: the captured step needs stable gradient addresses, so the models switch to zero-copy gradient hand-off
        # (rvt_amd/backbone.py: .grad = persistent bucket views, torch.autograd.grad sees no backbone gradients); close()
        # restores the flags for later eager use of the same modules
        self._saved_flags = [getattr(m, 'zero_copy_grads', False) for m in self.models]
        for m in self.models:
            m.zero_copy_grads = True
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        for _ in range(max(1, warmup)):          # eager warm-up: sizes every grow-only workspace, the occupancy caches and
            fn()                                 # the optimizer state before anything is recorded
        torch.cuda.synchronize(dev)
        # the capture allocates the step's activations once more, in the graph's private pool: give the eager steps'
        # cached blocks back first (RVT-Base 1 Mpx T=21 B=24 keeps ~110 GiB cached per pool)
        torch.cuda.empty_cache()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()

    def close(self) -> None:
        """Drop the graph and give the models their gradient hand-off mode back (eager use after a captured phase)."""
        self.graph = None
        for m, f in zip(self.models, self._saved_flags):
            m.zero_copy_grads = f
            for p in m.parameters():
                p.grad = None            # bucket views would be overwritten by the next eager backward
            m.invalidate_weight_cache()

    def __call__(self) -> None:
        self.graph.replay()
        for m in self.models:
            m.invalidate_weight_cache()
