"""The training / evaluation step around the backbone — the reference's Lightning ``Module``
(modules/detection.py:104-280) restated without Lightning (not installed here), keeping the
``training_step(batch, batch_idx) -> {'loss': …}`` / ``validation_step`` signatures and the batch dict
contract so the object can stand in for it under a Trainer.

What changes is the schedule, not the semantics: the reference calls the backbone T times and chains
states through autograd (modules/detection.py:131-148); here the whole list of T event tensors goes
through ``RNNDetector.forward_sequence`` once (stage-major, rvt_amd/stage.py).  The features of labelled
frames are then gathered exactly like ``BackboneFeatureSelector`` (modules/utils/detection.py:24-46) and
handed to the detection head, which stays whatever the caller provides (the reference's YOLOX
FPN/head through ``forward_detect``) — it is outside this package's scope.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch

from . import ops
from .states import RNNStates, merge_mixed_batches
from .types import DataType, Mode


class BackboneSequenceModule:
    """``detect_fn(backbone_features: {stage: (N,C,H,W)}, labels) -> {'loss': tensor, …}`` is the head+loss."""

    def __init__(self, backbone, detect_fn: Callable, in_stages=(2, 3, 4)):
        self.backbone = backbone
        self.detect_fn = detect_fn
        self.in_stages = tuple(in_stages)
        self.mode_2_rnn_states = {m: RNNStates() for m in Mode}
        self.mode_2_batch_size = {m: None for m in Mode}

    # -- shared time-loop replacement ----------------------------------------------------------------------
    def _run_sequence(self, data, worker_id: int, mode: Mode):
        ev_seq = data[DataType.EV_REPR]                     # list of T tensors (B,C,h,w), uint8 or float
        labels_seq = data[DataType.OBJLABELS_SEQ]           # list of T label containers (len == B)
        is_first = data[DataType.IS_FIRST_SAMPLE]
        token_masks = data.get(DataType.TOKEN_MASK, None)
        states = self.mode_2_rnn_states[mode]
        states.reset(worker_id=worker_id, indices_or_bool_tensor=is_first)          # detection.py:117
        T = len(ev_seq)
        assert T > 0
        B = len(labels_seq[0])
        if self.mode_2_batch_size[mode] is None:
            self.mode_2_batch_size[mode] = B
        else:
            assert self.mode_2_batch_size[mode] == B
        prev = states.get_states(worker_id=worker_id)
        tm = None if token_masks is None else torch.stack(list(token_masks), 0)
        feats, new_states = self.backbone.forward_sequence(ev_seq, prev, tm)        # replaces detection.py:131-148
        states.save_states_and_detach(worker_id=worker_id, states=new_states)       # detection.py:159
        return feats, labels_seq, T, B

    @staticmethod
    def _select_labelled(feats, labels_seq, T, stages):
        """BackboneFeatureSelector semantics (modules/utils/detection.py:32-46): concatenate, over t, the batch rows that
        carry labels — as ONE device gather per stage over the flattened (t, b) frame axis (rvt_gather_frames) instead of
        T index_selects and a cat."""
        flat_idx: List[int] = []
        labels = []
        B = feats[stages[0]].shape[1]
        for t in range(T):
            cur, valid_idx = labels_seq[t].get_valid_labels_and_batch_indices()
            if len(cur) > 0:
                flat_idx.extend(t * B + int(b) for b in valid_idx)
                labels.extend(cur)
        if not labels:
            return None, labels
        dev = feats[stages[0]].device
        idx = torch.tensor(flat_idx, dtype=torch.int32).to(dev)
        sel = {}
        for s in stages:
            f = feats[s]                                         # (T, B, C, H, W)-shaped view of channels-last storage
            cl = f.permute(0, 1, 3, 4, 2)                        # (T, B, H, W, C): contiguous for the backbone's outputs
            if not cl.is_contiguous():
                cl = cl.contiguous()
            g = ops.gather_frames(cl.reshape(cl.shape[0] * cl.shape[1], *cl.shape[2:]), idx)
            sel[s] = g.permute(0, 3, 1, 2)                       # back to NCHW-shaped (channels-last strides), as the FPN expects
        return sel, labels

    # -- Lightning-compatible entry points ------------------------------------------------------------------
    def training_step(self, batch: Any, batch_idx: int) -> Dict[str, Any]:
        batch = merge_mixed_batches(batch)
        feats, labels_seq, T, B = self._run_sequence(batch['data'], batch['worker_id'], Mode.TRAIN)
        sel, labels = self._select_labelled(feats, labels_seq, T, self.in_stages)
        assert len(labels) > 0
        out = self.detect_fn(sel, labels)
        assert 'loss' in out
        return out

    def _val_test_step_impl(self, batch: Any, mode: Mode) -> Optional[Dict[str, Any]]:
        with torch.no_grad():
            feats, labels_seq, T, B = self._run_sequence(batch['data'], batch['worker_id'], mode)
            sel, labels = self._select_labelled(feats, labels_seq, T, self.in_stages)
            if not labels:
                return {'skip': True}
            return self.detect_fn(sel, labels)

    def validation_step(self, batch: Any, batch_idx: int):
        return self._val_test_step_impl(batch, Mode.VAL)

    def test_step(self, batch: Any, batch_idx: int):
        return self._val_test_step_impl(batch, Mode.TEST)
