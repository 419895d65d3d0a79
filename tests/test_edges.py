"""Edge cases of the backbone against the CPU oracle (fp32; emulator and GPU): shapes and inputs the golden cases do not
hit — a single frame, a single sample, an odd batch, an all-zero event tensor, saturated (255) counts, every token masked,
no token masked."""
import numpy as np
import pytest
import torch

from oracle import rvt_oracle as O
from tests import casegen
from tests.backends import backend  # noqa: F401
from tests.test_backbone import build_model

EDGE = {
    # name: (base case, T, B, input kind, mask kind)
    'one_frame_one_sample': ('micro', 1, 1, 'rand', None),
    'odd_batch': ('micro', 2, 3, 'rand', None),
    'all_zero_events': ('micro', 2, 2, 'zeros', None),
    'saturated_events': ('micro', 2, 1, 'max', None),
    'all_tokens_masked': ('micro_mask', 2, 2, 'rand', 'all'),
    'no_token_masked': ('micro_mask', 2, 2, 'rand', 'none'),
}


def _inputs(case, T, B, kind):
    c = casegen.CASES[case]
    h, w = c['hw']
    if kind == 'zeros':
        return torch.zeros(T, B, 20, h, w, dtype=torch.uint8)
    if kind == 'max':
        return torch.full((T, B, 20, h, w), 255, dtype=torch.uint8)
    g = torch.Generator().manual_seed(11)
    return torch.randint(0, 11, (T, B, 20, h, w), generator=g, dtype=torch.uint8)


@pytest.mark.parametrize('name', list(EDGE))
def test_edge_case_matches_oracle(backend, name):
    case, T, B, kind, mkind = EDGE[name]
    c, cfgd = casegen.CASES[case], casegen.case_cfg(case)
    xs = _inputs(case, T, B, kind)
    Hs, Ws = c['in_res'][0] // 4, c['in_res'][1] // 4
    masks = None
    if mkind is not None:
        masks = torch.ones(T, B, Hs, Ws, dtype=torch.bool) if mkind == 'all' else torch.zeros(T, B, Hs, Ws, dtype=torch.bool)
    # oracle
    cfg = O.OracleCfg(**{k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in cfgd.items()})
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in casegen.make_params(cfgd, seed=0, gamma=c['gamma']).items()}
    feats_o, states_o = O.sequence_forward(xs, None, params, cfg, c['in_res'], torch.float32, masks)
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(feats_o[0][s + 1].shape, generator=g) for s in range(4)]
    loss_o = sum((feats_o[t][s + 1] * cots[s]).sum() for t in range(T) for s in range(4))
    grads_o = dict(zip(params.keys(), torch.autograd.grad(loss_o, list(params.values()))))
    # HIP path
    m = build_model(case, backend, torch.float32)
    feats, states = m.forward_sequence(xs.to(backend), None, None if masks is None else masks.to(backend))
    loss = sum((feats[s + 1][t].float() * cots[s].to(backend)).sum() for t in range(T) for s in range(4))
    loss.backward()

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    for s in range(4):
        for t in range(T):
            assert rel(feats[s + 1][t], feats_o[t][s + 1]) <= 1e-3, (name, 'feat', s, t)
        assert rel(states[s][1], states_o[s][1]) <= 1e-3, (name, 'cell', s)
    for k, p in m.named_parameters():
        if grads_o[k].abs().max() == 0:                       # e.g. mask token when nothing is masked
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, (name, k)
        else:
            assert rel(p.grad, grads_o[k]) <= 1e-3, (name, 'grad', k, rel(p.grad, grads_o[k]))
