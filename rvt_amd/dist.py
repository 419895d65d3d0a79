"""Data-parallel gradient exchange for the backbone: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference leaves this to Lightning's DDPStrategy (train.py:60-67): one reducer over the whole
detector whose buckets only become ready when autograd reaches t=0 of the time loop, i.e. at the very
end of backward.  The stage-major backward (rvt_amd/stage.py) finishes stage 4 — 75 % of the backbone's
parameter bytes — first, and every stage's parameter gradients already live in ONE persistent flat fp32
bucket (rvt_amd/weights.py: the weight-gradient kernels accumulate straight into views of it), so the
moment a stage's backward returns its bucket is all-reduced in place, asynchronously, overlapping stages 3..1.
No flattening copy, no per-step allocation.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): four large
buckets (38 / 9.6 / 2.4 / 0.8 MB fp32 for RVT-Base) keep every collective bandwidth- rather than latency-bound.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class StageGradReducer:
    """Attach with ``reducer.attach(model)``.  The backbone's backward waits for the collectives itself before it
    hands the gradients (views of the buckets) to the parameters; ``reducer.finish()`` after ``loss.backward()`` is
    harmless (idempotent) and kept for callers that drive the hook by hand."""

    def __init__(self, process_group=None, average: bool = True, force: bool = False):
        self.pg = process_group
        self.average = average
        self.force = force          # run the collective even at world_size 1 (single-GPU smoke of the RCCL path)
        self._pending: List = []
        # observability of the overlap claim (bench.py --gpus N): HIP events on the compute stream at the launch of the LAST bucket
        # of a backward and behind the waits of finish(); their distance is the communication the backward did NOT hide
        self.record_tail = False
        self._tail_events: List = []
        self._ev_last_launch = None

    def attach(self, model) -> 'StageGradReducer':
        model._stage_grad_hook = self.on_stage_done
        # the backbone calls this at the very end of its backward, before the bucket views become the parameters'
        # .grad: whatever reads them next (the optimizer) is then stream-ordered after the collectives, while the
        # collectives themselves still overlap the backward of the later stages
        model._stage_grad_finish = self.finish
        return self

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.pg) if dist.is_initialized() else 1

    def on_stage_done(self, stage_idx: int, bucket: torch.Tensor, accumulated: bool = False) -> None:
        """All-reduce (mean) the flat fp32 gradient bucket of one stage in place.  Async: the collective waits for the
        producing stream, the consumer waits in finish().  `accumulated`: the bucket still holds the (already reduced)
        gradients of earlier backward passes (zero-copy hand-off without zeroing): averaging keeps them intact
        (mean(g_prev + g_i) = g_prev + mean(g_i)), a SUM would multiply them by the world size."""
        ws = self.world_size
        if accumulated and not self.average and ws > 1:
            raise RuntimeError('StageGradReducer(average=False) with gradient accumulation into the persistent buckets: '
                               'zero the gradients (set_to_none=True) before every backward, or use average=True')
        if (ws == 1 and not self.force) or bucket.numel() == 0:
            return
        if not dist.is_initialized():
            return
        op = dist.ReduceOp.SUM
        if self.average and ws > 1:
            if dist.get_backend(self.pg) == 'nccl':
                op = dist.ReduceOp.AVG              # RCCL divides in the collective: no extra pass over the bucket
            else:
                bucket.div_(ws)
        self._pending.append(dist.all_reduce(bucket, op=op, group=self.pg, async_op=True))
        if self.record_tail and bucket.is_cuda:
            self._ev_last_launch = torch.cuda.Event(enable_timing=True)
            self._ev_last_launch.record()

    def finish(self) -> None:
        had = bool(self._pending)
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if had and self.record_tail and self._ev_last_launch is not None:
            done = torch.cuda.Event(enable_timing=True)
            done.record()
            self._tail_events.append((self._ev_last_launch, done))
            self._ev_last_launch = None

    def tail_ms(self) -> List[float]:
        """Per backward since the last call: GPU time between the launch of the last stage bucket's all-reduce (in stream order:
        the end of the backbone backward) and the point where every collective of that backward has completed."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._tail_events]
        self._tail_events.clear()
        return out
