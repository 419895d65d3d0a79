"""Shared test harness: run an implementation on a casegen case and reduce its results to the
same keys oracle/make_golden.py stores, so that (reference golden) vs (oracle) vs (HIP path)
comparisons all go through one code path."""
from __future__ import annotations

import os
from typing import Callable, Dict, List

import numpy as np
import torch

from tests import casegen

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name: str) -> Dict[str, np.ndarray]:
    with np.load(os.path.join(GOLDEN_DIR, f'{name}.npz')) as z:
        return {k: z[k] for k in z.files}


def summarize(name: str, feats_all: List[Dict[int, torch.Tensor]], states, grads: Dict[str, torch.Tensor] = None,
              prefix: str = '') -> Dict[str, np.ndarray]:
    """feats_all[t][stage] NCHW-shaped tensors, states [(h,c)]*4 NCHW-shaped, grads by reference param name."""
    T = len(feats_all)
    out = {}
    for s in range(4):
        f_last = feats_all[T - 1][s + 1].detach().float().cpu().contiguous().numpy().reshape(-1)
        idx = casegen.sample_idx(f_last.size, 4096)
        out[f'{prefix}feat{s}_last_samples'] = f_last[idx]
        if prefix:
            out[f'{prefix}feat{s}_sums'] = np.array([float(feats_all[T - 1][s + 1].detach().double().sum()),
                                                     float(feats_all[T - 1][s + 1].detach().double().abs().sum())])
        else:
            out[f'feat{s}_sums'] = np.array([[float(feats_all[t][s + 1].detach().double().sum()),
                                              float(feats_all[t][s + 1].detach().double().abs().sum())] for t in range(T)])
            c_last = states[s][1].detach().float().cpu().contiguous().numpy().reshape(-1)
            out[f'cell{s}_last_samples'] = c_last[idx]
            out[f'cell{s}_sums'] = np.array([float(states[s][1].detach().double().sum()), float(states[s][1].detach().double().abs().sum())])
            if name in casegen.FULL_TENSOR_CASES:      # compared element for element over the whole tensor
                out[f'feat{s}_last_full'] = feats_all[T - 1][s + 1].detach().float().cpu().contiguous().numpy()
                out[f'cell{s}_last_full'] = states[s][1].detach().float().cpu().contiguous().numpy()
    if grads is not None:
        for k, g in grads.items():
            g = g.detach().double().cpu().contiguous().numpy().reshape(-1)
            out[f'grad/{k}/stats'] = np.array([g.sum(), np.sqrt((g * g).sum())])
            out[f'grad/{k}/samples'] = g[casegen.sample_idx(g.size, 256)].astype(np.float32)
    return out


def compare(got: Dict[str, np.ndarray], want: Dict[str, np.ndarray], rtol: float, what: str,
            grad_rtol: float = None, keys_prefix: str = None):
    """Relative-to-scale comparison: |got-want|_max <= rtol * max(|want|_max, tiny) per key;
    sums compared relative to the abs-sum.  Returns the worst ratio seen (for reporting)."""
    grad_rtol = grad_rtol or rtol
    worst = 0.0
    for k, w in want.items():
        if k == 'loss' or k not in got:
            continue
        if keys_prefix is not None and not k.startswith(keys_prefix):
            continue
        g = got[k]
        tol = grad_rtol if k.startswith('grad/') else rtol
        if k.endswith('_sums'):
            w2, g2 = w.reshape(-1, 2), g.reshape(-1, 2)
            err = np.abs(g2 - w2).max(axis=None) / max(np.abs(w2[:, 1]).max(), 1e-30)
        elif k.endswith('/stats'):
            err = abs(g[1] - w[1]) / max(abs(w[1]), 1e-30)          # l2 norm
        else:
            scale = max(np.abs(w).max(), 1e-30)
            err = np.abs(g.astype(np.float64) - w.astype(np.float64)).max() / scale
        worst = max(worst, err / tol)
        assert err <= tol, f'{what}: {k}: rel err {err:.3e} > {tol:.1e}'
    return worst


def oracle_run(name: str, dtype=torch.float32, with_grads=True, with_batch2=True, conv_impl: str = 'im2col'):
    """Run the CPU oracle on a case; returns the summarized dict."""
    from oracle import rvt_oracle as O
    c = casegen.CASES[name]
    cfgd = casegen.case_cfg(name)
    cfg = O.OracleCfg(**{k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in cfgd.items()}, conv_impl=conv_impl)
    params = {k: torch.from_numpy(v).to(dtype).requires_grad_(True)
              for k, v in casegen.make_params(cfgd, seed=0, gamma=c['gamma']).items()}
    xs = torch.from_numpy(casegen.make_inputs(name))
    masks = torch.from_numpy(casegen.make_token_masks(name)) if cfgd['enable_masking'] else None
    cots = [torch.from_numpy(a).to(dtype) for a in casegen.make_cotangents(name)]
    try:
        feats_all, states = O.sequence_forward(xs, None, params, cfg, c['in_res'], dtype, masks)
    finally:
        O._ATEN[0] = False
    grads = None
    if with_grads:
        loss = sum((feats_all[t][s + 1] * cots[s][t]).sum() for t in range(c['T']) for s in range(4))
        gl = torch.autograd.grad(loss, list(params.values()))
        grads = dict(zip(params.keys(), gl))
    out = summarize(name, feats_all, states, grads)
    if with_batch2:
        with torch.no_grad():
            first = torch.tensor([True] + [False] * (c['B'] - 1))
            st2 = O.reset_states([(h.detach(), cc.detach()) for h, cc in states], first)
            feats2, _ = O.sequence_forward(xs, st2, {k: v.detach() for k, v in params.items()}, cfg,
                                           c['in_res'], dtype, masks)
        out.update(summarize(name, feats2, None, None, prefix='b2_'))
    return out
