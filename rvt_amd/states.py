"""Recurrent-state bookkeeping around the backbone (reference modules/utils/detection.py:76-161).

Same behaviour as the reference's ``RNNStates`` (dict worker_id -> [(h,c)]*4, ``None`` until the first
save, detach on save, masked in-place reset) with the reset done by the HIP kernel
``rvt_state_reset_masked`` instead of advanced-indexing assignment."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Union

import torch

from . import ops
from .types import DatasetSamplingMode, LstmStates


class RNNStates:
    def __init__(self):
        self.states: Dict[int, LstmStates] = {}

    def _has_states(self) -> bool:
        return len(self.states) > 0

    @classmethod
    def recursive_detach(cls, inp):
        if isinstance(inp, torch.Tensor):
            d = inp.detach()
            # a state that is a slice of a larger buffer (the (T+1)-slot feature array of the sequence kernels) would
            # pin that whole buffer for as long as the state is kept: keep a copy of the slice instead
            base = d._base
            if base is not None and base.numel() > d.numel():
                d = d.clone(memory_format=torch.preserve_format)
            return d
        if isinstance(inp, list):
            return [cls.recursive_detach(x) for x in inp]
        if isinstance(inp, tuple):
            return tuple(cls.recursive_detach(x) for x in inp)
        if isinstance(inp, dict):
            return {k: cls.recursive_detach(v) for k, v in inp.items()}
        raise NotImplementedError

    @classmethod
    def recursive_reset(cls, inp, indices_or_bool_tensor: Optional[Union[List[int], torch.Tensor]] = None):
        if isinstance(inp, torch.Tensor):
            assert inp.requires_grad is False, 'Not assumed here but should be the case.'
            B = inp.shape[0]
            if indices_or_bool_tensor is None:
                mask = torch.ones(B, dtype=torch.bool)
            else:
                assert len(indices_or_bool_tensor) > 0
                sel = torch.as_tensor(indices_or_bool_tensor)
                if sel.dtype == torch.bool:
                    mask = sel
                else:
                    mask = torch.zeros(B, dtype=torch.bool)
                    mask[sel.long().cpu()] = True
            # states are NCHW-shaped views of channels-last buffers: reset the underlying storage order
            base = inp.permute(0, 2, 3, 1) if inp.dim() == 4 and not inp.is_contiguous() else inp
            if base.is_contiguous() and base.dtype in (torch.float32, torch.bfloat16) and (base.is_cuda or ops.L.is_emulator()):
                ops.state_reset_masked(base, mask)
            else:                       # foreign tensors (e.g. CPU states from a checkpoint): plain indexing
                inp[mask.to(inp.device)] = 0
            return inp
        if isinstance(inp, list):
            return [cls.recursive_reset(x, indices_or_bool_tensor) for x in inp]
        if isinstance(inp, tuple):
            return tuple(cls.recursive_reset(x, indices_or_bool_tensor) for x in inp)
        if isinstance(inp, dict):
            return {k: cls.recursive_reset(v, indices_or_bool_tensor) for k, v in inp.items()}
        raise NotImplementedError

    def save_states_and_detach(self, worker_id: int, states: LstmStates) -> None:
        self.states[worker_id] = self.recursive_detach(states)

    def get_states(self, worker_id: int) -> Optional[LstmStates]:
        if not self._has_states() or worker_id not in self.states:
            return None
        return self.states[worker_id]

    def reset(self, worker_id: int, indices_or_bool_tensor=None):
        if not self._has_states():
            return
        if worker_id in self.states:
            self.states[worker_id] = self.recursive_reset(self.states[worker_id], indices_or_bool_tensor)


def mixed_collate_fn(x1, x2):
    """reference modules/utils/detection.py:133-144"""
    if isinstance(x1, torch.Tensor):
        assert isinstance(x2, torch.Tensor)
        return torch.cat((x1, x2))
    if isinstance(x1, list):
        assert isinstance(x2, list) and len(x1) == len(x2)
        return [mixed_collate_fn(a, b) for a, b in zip(x1, x2)]
    if hasattr(x1, '__add__'):          # SparselyBatchedObjectLabels
        return x1 + x2
    raise NotImplementedError


def merge_mixed_batches(batch: Dict[str, Any]):
    """reference modules/utils/detection.py:147-161"""
    if 'data' in batch:
        return batch
    rnd_data = batch[DatasetSamplingMode.RANDOM]['data']
    stream_batch = batch[DatasetSamplingMode.STREAM]
    out = {'worker_id': stream_batch['worker_id']}
    stream_data = stream_batch['data']
    assert rnd_data.keys() == stream_data.keys()
    out['data'] = {k: mixed_collate_fn(stream_data[k], rnd_data[k]) for k in rnd_data.keys()}
    return out
