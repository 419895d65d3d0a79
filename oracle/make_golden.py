"""Generate golden fixtures from the UNMODIFIED reference, imported from /root/reference.

TEST INFRASTRUCTURE.  Runs only in the authoring container (the reference tree does not
exist on the GPU box).  Usage:  python oracle/make_golden.py [case ...]

For every case in tests/casegen.py:CASES it
  1. builds the reference ``RNNDetector`` through ``build_recurrent_backbone`` with the
     reference's own config keys (config/model/maxvit_yolox/default.yaml),
  2. asserts its ``state_dict()`` names/shapes equal tests/casegen.py:param_shapes (the
     checkpoint-compat contract),
  3. loads the numpy-seeded parameters, runs the time loop exactly like
     modules/detection.py:131-148 (cast → zero-pad → forward, chaining states),
  4. back-propagates  L = Σ_t Σ_s <feat[t][s], cot[s][t]>  and
  5. stores sampled outputs / states / parameter gradients in tests/golden/<case>.npz.

Nothing from the reference's source is copied; only its numerical outputs are recorded.
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, '_stubs'))
sys.path.insert(1, '/root/reference')
sys.path.insert(2, ROOT)

import numpy as np
import torch
import torch.nn.functional as F

from tests import casegen  # noqa: E402


def reference_cfg(cfg: dict):
    from omegaconf import OmegaConf
    return OmegaConf.create({
        'name': 'MaxViTRNN',
        'compile': {'enable': False, 'args': {'mode': 'reduce-overhead'}},
        'input_channels': cfg['input_channels'],
        'enable_masking': cfg['enable_masking'],
        'partition_split_32': 1,
        'embed_dim': cfg['embed_dim'],
        'dim_multiplier': list(cfg['dim_multiplier']),
        'num_blocks': list(cfg['num_blocks']),
        'T_max_chrono_init': [4, 8, 16, 32],
        'stem': {'patch_size': cfg['patch_size']},
        'stage': {
            'downsample': {'type': 'patch', 'overlap': cfg['overlap'], 'norm_affine': True},
            'attention': {'use_torch_mha': False, 'partition_size': list(cfg['partition_size']),
                          'dim_head': cfg['dim_head'], 'attention_bias': True, 'mlp_activation': 'gelu',
                          'mlp_gated': False, 'mlp_bias': True, 'mlp_ratio': 4, 'drop_mlp': 0, 'drop_path': 0,
                          'ls_init_value': 1e-5},
            'lstm': {'dws_conv': cfg['dws_conv'], 'dws_conv_only_hidden': cfg['dws_conv_only_hidden'],
                     'dws_conv_kernel_size': cfg['dws_conv_kernel_size'], 'drop_cell_update': 0},
        },
    })


def build_reference(cfg: dict):
    from models.detection.recurrent_backbone import build_recurrent_backbone
    m = build_recurrent_backbone(reference_cfg(cfg))
    sd = m.state_dict()
    want = casegen.param_shapes(cfg)
    got = [(k, tuple(v.shape)) for k, v in sd.items()]
    assert sorted(got) == sorted(want), (set(got) ^ set(want))
    return m


def run_case(name: str, out_dir: str):
    c = casegen.CASES[name]
    cfg = casegen.case_cfg(name)
    torch.manual_seed(0)
    m = build_reference(cfg).float()
    params = casegen.make_params(cfg, seed=0, gamma=c['gamma'])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m.train()  # no dropout/BN in the backbone; keeps autograd on

    xs = torch.from_numpy(casegen.make_inputs(name))
    cots = [torch.from_numpy(a) for a in casegen.make_cotangents(name)]
    masks = torch.from_numpy(casegen.make_token_masks(name)) if cfg['enable_masking'] else None
    T = c['T']
    H, W = c['in_res']

    def run_sequence(states):
        feats_all = []
        for t in range(T):
            x = xs[t].to(torch.float32)
            x = F.pad(x, [0, W - x.shape[-1], 0, H - x.shape[-2]])
            feats, states = m(x, states, None if masks is None else masks[t])
            feats_all.append(feats)
        return feats_all, states

    feats_all, states = run_sequence(None)
    loss = sum((feats_all[t][s + 1] * cots[s][t]).sum() for t in range(T) for s in range(4))
    loss.backward()

    out = {'loss': np.float64(loss.item())}
    for s in range(4):
        f_last = feats_all[T - 1][s + 1].detach().numpy().reshape(-1)
        idx = casegen.sample_idx(f_last.size, 4096)
        out[f'feat{s}_last_samples'] = f_last[idx]
        out[f'feat{s}_sums'] = np.array([[feats_all[t][s + 1].sum().item(), feats_all[t][s + 1].abs().sum().item()]
                                         for t in range(T)], dtype=np.float64)
        c_last = states[s][1].detach().numpy().reshape(-1)
        out[f'cell{s}_last_samples'] = c_last[idx]
        out[f'cell{s}_sums'] = np.array([states[s][1].sum().item(), states[s][1].abs().sum().item()])
        if name in casegen.FULL_TENSOR_CASES:          # whole tensors, not samples (NCHW as the reference returns them)
            out[f'feat{s}_last_full'] = feats_all[T - 1][s + 1].detach().numpy().copy()
            out[f'cell{s}_last_full'] = states[s][1].detach().numpy().copy()
    for k, prm in m.named_parameters():
        g = prm.grad.detach().numpy().reshape(-1).astype(np.float64)
        out[f'grad/{k}/stats'] = np.array([g.sum(), np.sqrt((g * g).sum())])
        out[f'grad/{k}/samples'] = g[casegen.sample_idx(g.size, 256)].astype(np.float32)

    # second batch: carry detached states, reset sample 0 (modules/utils/detection.py:96-130)
    with torch.no_grad():
        st2 = [(h.detach().clone(), cc.detach().clone()) for h, cc in states]
        for h, cc in st2:
            h[torch.tensor([True] + [False] * (c['B'] - 1))] = 0
            cc[torch.tensor([True] + [False] * (c['B'] - 1))] = 0
        feats2, states2 = run_sequence(st2)
        for s in range(4):
            f_last = feats2[T - 1][s + 1].numpy().reshape(-1)
            out[f'b2_feat{s}_last_samples'] = f_last[casegen.sample_idx(f_last.size, 4096)]
            out[f'b2_feat{s}_sums'] = np.array([feats2[T - 1][s + 1].sum().item(),
                                                feats2[T - 1][s + 1].abs().sum().item()])

    path = os.path.join(out_dir, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: loss={loss.item():.6f}  -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count() or 1)
    names = sys.argv[1:] or list(casegen.CASES)
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    for n in names:
        run_case(n, out_dir)
    assert not os.path.exists('/root/reference/models/__pycache__'), 'bytecode leaked into the reference tree'
