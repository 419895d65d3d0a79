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


class GraphedStreamStep:
    """One streaming-inference step (T = 1, the ConvLSTM state carried from call to call: the validation / deployment loop of
    modules/detection.py:231-255) as ONE hipGraph launch.  A step is ~50 short launches (20 - 250 us each at B = 64); replayed as a
    graph the gaps between them shrink.  The input frame and the recurrent state live in static buffers:

        stream = GraphedStreamStep(model, example_frame)       # eager warm-up + capture (the state starts from zeros)
        feats = stream(frame)                                   # copies `frame` in, replays; feats: {stage: (B, C, H, W)} STATIC tensors
        stream.reset()  /  stream.reset(mask)                   # new sequences: zero the state (of the samples where mask is set)

    The features returned are the graph's own buffers: valid until the next call (copy what has to outlive it).  The parameters must
    not change while the graph lives (inference); call close() before training the model again."""

    def __init__(self, model, example_frame: torch.Tensor, warmup: int = 3):
        assert example_frame.is_cuda and not model.training, 'GraphedStreamStep: an eval-mode model and a CUDA frame (B, C, h, w)'
        self.model = model
        self.x = torch.empty_like(example_frame)
        self.x.copy_(example_frame)
        with torch.no_grad():
            st = None
            for _ in range(max(2, warmup)):                     # eager warm-up: weight pack, workspaces, occupancy caches
                _, st = model(self.x, st)
            # static state buffers (same shapes / strides as the states the model hands back)
            self.states = [tuple(t.clone(memory_format=torch.preserve_format) for t in pair) for pair in st]
            self.reset()
            torch.cuda.synchronize(self.x.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                feats, new = model(self.x, self.states)
                for (h, c), (hn, cn) in zip(self.states, new):
                    h.copy_(hn)
                    c.copy_(cn)
            self.feats = feats
        self.reset()

    def reset(self, mask: Optional[torch.Tensor] = None) -> None:
        """Zero the recurrent state - of every sample, or of those where the bool mask (B,) is set (modules/utils/detection.py:96-113)."""
        for h, c in self.states:
            if mask is None:
                h.zero_()
                c.zero_()
            else:
                h[mask] = 0
                c[mask] = 0

    def __call__(self, frame: torch.Tensor):
        self.x.copy_(frame)
        self.graph.replay()
        return self.feats

    def close(self) -> None:
        self.graph = None
        self.feats = None
