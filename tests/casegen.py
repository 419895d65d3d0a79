"""Deterministic (numpy-seeded) parameter / input / cotangent generators shared by
``oracle/make_golden.py`` (run in the authoring container, next to the reference) and the
parity tests (run anywhere, including the GPU box where the reference does not exist).

Everything is derived from ``numpy.random.default_rng(seed)`` streams keyed by name, so the
golden fixtures only have to store *outputs*, never weights or inputs.

The parameter name/shape table restates the reference ``state_dict()`` contract (SURVEY.md §8b;
reference: maxvit_rnn.py:130-167, maxvit.py:147-172,193-250,328-341, rnn.py:11-34);
``oracle/make_golden.py`` asserts it equals the real reference module's state dict.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

CASES: Dict[str, dict] = {
    # name: cfg overrides + workload.  hw = raw sensor resolution, in_res = padded model resolution
    'micro': dict(embed_dim=16, dim_head=16, partition_size=(2, 3), hw=(60, 90), in_res=(64, 96),
                  T=3, B=2, gamma='rand'),
    'micro_default_gamma': dict(embed_dim=16, dim_head=16, partition_size=(2, 3), hw=(60, 90), in_res=(64, 96),
                                T=3, B=2, gamma='default'),
    'micro_dh24': dict(embed_dim=48, dim_head=24, partition_size=(2, 3), hw=(64, 96), in_res=(64, 96),
                       T=2, B=2, gamma='rand'),
    'micro_dws_hidden': dict(embed_dim=16, dim_head=16, partition_size=(2, 3), hw=(64, 96), in_res=(64, 96),
                             T=3, B=2, gamma='rand', dws_conv=True, dws_conv_only_hidden=True),
    'micro_dws_xh': dict(embed_dim=16, dim_head=16, partition_size=(2, 3), hw=(64, 96), in_res=(64, 96),
                         T=3, B=2, gamma='rand', dws_conv=True, dws_conv_only_hidden=False),
    'micro_mask': dict(embed_dim=16, dim_head=16, partition_size=(2, 3), hw=(64, 96), in_res=(64, 96),
                       T=2, B=2, gamma='rand', enable_masking=True),
    'micro_nooverlap': dict(embed_dim=16, dim_head=16, partition_size=(2, 3), hw=(64, 96), in_res=(64, 96),
                            T=2, B=2, gamma='rand', overlap=False),
    # BASELINE.json configs[0]: RVT-Tiny, Gen1 shape, T=5, batch=2 (CPU plumbing case)
    'tiny_gen1': dict(embed_dim=32, dim_head=32, partition_size=(8, 10), hw=(240, 304), in_res=(256, 320),
                      T=5, B=2, gamma='default'),
    'tiny_gen1_gamma': dict(embed_dim=32, dim_head=32, partition_size=(8, 10), hw=(240, 304), in_res=(256, 320),
                            T=5, B=2, gamma='rand'),
    # RVT-Base widths (stage 1: C = 64, two heads, 6x10 partitions = the fused attention-half kernels) at a resolution the
    # CPU emulator of the kernel sources gets through: 48x80 tokens at stage 1, one partition per frame at stage 4
    'base_qvga': dict(embed_dim=64, dim_head=32, partition_size=(6, 10), hw=(180, 300), in_res=(192, 320),
                      T=2, B=1, gamma='rand'),
    # RVT-Base on the 1Mpx shape, reduced B*T so that the CPU oracle finishes in seconds
    'base_1mpx': dict(embed_dim=64, dim_head=32, partition_size=(6, 10), hw=(360, 640), in_res=(384, 640),
                      T=2, B=1, gamma='rand'),
    # ---- round 3: the depth every BASELINE GPU config runs at (T = 21, config/experiment/gen1/default.yaml:37) ----
    # BASELINE configs[2] per sample: RVT-Base, 1Mpx, the full 21-step recurrence (B = 1 so that the CPU reference needs seconds)
    'base_1mpx_t21': dict(embed_dim=64, dim_head=32, partition_size=(6, 10), hw=(360, 640), in_res=(384, 640),
                          T=21, B=1, gamma='rand'),
    # BASELINE configs[1] at reduced batch: RVT-Tiny, Gen1, T = 21
    'tiny_gen1_t21': dict(embed_dim=32, dim_head=32, partition_size=(8, 10), hw=(240, 304), in_res=(256, 320),
                          T=21, B=2, gamma='rand'),
    # RVT-Base on Gen1 (config/experiment/gen1/base.yaml): C = 64 with the 8 x 10 = 80-token partitions (three 32-token blocks
    # per partition in the fused attention-half kernels)
    'base_gen1': dict(embed_dim=64, dim_head=32, partition_size=(8, 10), hw=(240, 304), in_res=(256, 320),
                      T=2, B=2, gamma='rand'),
}

# cases whose fixtures additionally hold WHOLE tensors (last-step features and final cell states of every stage), not samples
FULL_TENSOR_CASES = ('micro', 'base_qvga')

CFG_DEFAULTS = dict(input_channels=20, dim_multiplier=(1, 2, 4, 8), num_blocks=(1, 1, 1, 1), patch_size=4,
                    overlap=True, norm_eps=1e-5, dws_conv=False, dws_conv_only_hidden=True,
                    dws_conv_kernel_size=3, enable_masking=False)


def case_cfg(name: str) -> dict:
    c = dict(CFG_DEFAULTS)
    for k, v in CASES[name].items():
        if k in ('hw', 'in_res', 'T', 'B', 'gamma'):
            continue
        c[k] = v
    return c


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in the reference's state_dict order."""
    out = []
    cin = cfg['input_channels']
    for si, mult in enumerate(cfg['dim_multiplier']):
        C = cfg['embed_dim'] * mult
        f = cfg['patch_size'] if si == 0 else 2
        k = (f - 1) * 2 + 1 if cfg['overlap'] else f
        pre = f'stages.{si}.'
        if cfg['enable_masking'] and si == 0:
            out.append((pre + 'mask_token', (1, 1, 1, C)))
        out.append((pre + 'downsample_cf2cl.conv.weight', (C, cin, k, k)))
        out.append((pre + 'downsample_cf2cl.norm.weight', (C,)))
        out.append((pre + 'downsample_cf2cl.norm.bias', (C,)))
        for bi in range(cfg['num_blocks'][si]):
            for blk in ('att_window', 'att_grid'):
                bp = f'{pre}att_blocks.{bi}.{blk}.'
                if not (bi == 0 and blk == 'att_window'):
                    out.append((bp + 'norm1.weight', (C,)))
                    out.append((bp + 'norm1.bias', (C,)))
                out.append((bp + 'self_attn.qkv.weight', (3 * C, C)))
                out.append((bp + 'self_attn.qkv.bias', (3 * C,)))
                out.append((bp + 'self_attn.proj.weight', (C, C)))
                out.append((bp + 'self_attn.proj.bias', (C,)))
                out.append((bp + 'ls1.gamma', (C,)))
                out.append((bp + 'norm2.weight', (C,)))
                out.append((bp + 'norm2.bias', (C,)))
                out.append((bp + 'mlp.net.0.0.weight', (4 * C, C)))
                out.append((bp + 'mlp.net.0.0.bias', (4 * C,)))
                out.append((bp + 'mlp.net.2.weight', (C, 4 * C)))
                out.append((bp + 'mlp.net.2.bias', (C,)))
                out.append((bp + 'ls2.gamma', (C,)))
        if cfg['dws_conv']:
            cg = C if cfg['dws_conv_only_hidden'] else 2 * C
            kk = cfg['dws_conv_kernel_size']
            out.append((pre + 'lstm.conv3x3_dws.weight', (cg, 1, kk, kk)))
            out.append((pre + 'lstm.conv3x3_dws.bias', (cg,)))
        out.append((pre + 'lstm.conv1x1.weight', (4 * C, 2 * C, 1, 1)))
        out.append((pre + 'lstm.conv1x1.bias', (4 * C,)))
        cin = C
    return out


def make_params(cfg: dict, seed: int, gamma: str) -> Dict[str, np.ndarray]:
    """fp32 parameters: weights ~ U(±1/sqrt(fan_in)) (PyTorch-default scale), LN weights ~ U(0.5,1.5),
    biases ~ U(±0.1); LayerScale γ = 1e-5 ('default', maxvit.py:49) or U(0.5,1.5) ('rand' — the variant
    that actually exercises attention/MLP, SURVEY.md §0 parity trap)."""
    p = {}
    for name, shape in param_shapes(cfg):
        r = _rng(seed, name)
        if name.endswith('gamma'):
            v = np.full(shape, 1e-5) if gamma == 'default' else r.uniform(0.5, 1.5, shape)
        elif name.endswith('norm.weight') or name.endswith('norm1.weight') or name.endswith('norm2.weight'):
            v = r.uniform(0.5, 1.5, shape)
        elif name.endswith('bias'):
            v = r.uniform(-0.1, 0.1, shape)
        elif name.endswith('mask_token'):
            v = r.normal(0.0, 0.02, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = 1.0 / np.sqrt(fan_in)
            v = r.uniform(-a, a, shape)
        p[name] = v.astype(np.float32)
    return p


def make_inputs(name: str, seed: int = 1) -> np.ndarray:
    """uint8 stacked-histogram-like event tensors (T,B,20,h,w), values 0..10
    (reference value range: data/utils/representations.py:62-67,117)."""
    c = CASES[name]
    r = _rng(seed, 'inputs/' + name)
    return r.integers(0, 11, size=(c['T'], c['B'], 20, *c['hw']), dtype=np.uint8)


def make_token_masks(name: str, seed: int = 1) -> np.ndarray:
    c = CASES[name]
    r = _rng(seed, 'mask/' + name)
    H, W = c['in_res']
    return r.random((c['T'], c['B'], H // 4, W // 4)) < 0.25


def stage_shapes(name: str) -> List[Tuple[int, int, int]]:
    c = CASES[name]
    cfg = case_cfg(name)
    H, W = c['in_res']
    return [(cfg['embed_dim'] * m, H // s, W // s) for m, s in zip(cfg['dim_multiplier'], (4, 8, 16, 32))]


def make_cotangents(name: str, seed: int = 3) -> List[np.ndarray]:
    """Upstream gradients for every stage output of every step: list over stages of (T,B,C,H,W) fp32.
    Stage 1 gets a cotangent too (the FPN ignores it but it exercises the full backward)."""
    c = CASES[name]
    out = []
    for si, (C, H, W) in enumerate(stage_shapes(name)):
        r = _rng(seed, f'cot/{name}/{si}')
        out.append(r.standard_normal((c['T'], c['B'], C, H, W)).astype(np.float32))
    return out


def sample_idx(n: int, k: int = 256) -> np.ndarray:
    if n <= k:
        return np.arange(n)
    return np.linspace(0, n - 1, k).astype(np.int64)
