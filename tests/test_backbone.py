"""End-to-end parity of the HIP backbone (forward, states, BPTT gradients of every parameter) against
(a) golden fixtures recorded from the unmodified reference and (b) the CPU oracle, on the emulator
build (small cases) and on the real GPU (all cases)."""
import numpy as np
import pytest
import torch

from rvt_amd import RNNDetector, AttrDict
from tests import casegen
from tests.backends import backend  # noqa: F401
from tests.harness import compare, load_golden, summarize

UNSUPPORTED = set()
EMU_CASES = ['micro', 'micro_default_gamma', 'micro_dh24', 'micro_mask', 'micro_nooverlap', 'micro_dws_hidden',
             'micro_dws_xh', 'base_qvga']
ALL_CASES = [c for c in casegen.CASES if c not in UNSUPPORTED]


def make_cfg(name):
    c = casegen.CASES[name]
    d = casegen.case_cfg(name)
    return AttrDict({
        'name': 'MaxViTRNN', 'input_channels': d['input_channels'], 'enable_masking': d['enable_masking'],
        'embed_dim': d['embed_dim'], 'dim_multiplier': list(d['dim_multiplier']), 'num_blocks': list(d['num_blocks']),
        'T_max_chrono_init': [4, 8, 16, 32], 'stem': {'patch_size': d['patch_size']}, 'in_res_hw': c['in_res'],
        'stage': {'downsample': {'type': 'patch', 'overlap': d['overlap'], 'norm_affine': True},
                  'attention': {'use_torch_mha': False, 'partition_size': d['partition_size'], 'dim_head': d['dim_head'],
                                'attention_bias': True, 'mlp_activation': 'gelu', 'mlp_gated': False, 'mlp_bias': True,
                                'mlp_ratio': 4, 'drop_mlp': 0, 'drop_path': 0, 'ls_init_value': 1e-5},
                  'lstm': {'dws_conv': d['dws_conv'], 'dws_conv_only_hidden': d['dws_conv_only_hidden'],
                           'dws_conv_kernel_size': d['dws_conv_kernel_size'], 'drop_cell_update': 0}}})


def build_model(name, device, dtype):
    c = casegen.CASES[name]
    m = RNNDetector(make_cfg(name), compute_dtype=dtype)
    sd = {k: torch.from_numpy(v) for k, v in casegen.make_params(casegen.case_cfg(name), seed=0, gamma=c['gamma']).items()}
    missing = m.load_state_dict(sd, strict=True)       # names/shapes == reference state_dict
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(device)


def run_hip_case(name, device, dtype, with_batch2=True):
    c = casegen.CASES[name]
    m = build_model(name, device, dtype)
    xs = torch.from_numpy(casegen.make_inputs(name)).to(device)
    masks = torch.from_numpy(casegen.make_token_masks(name)).to(device) if casegen.case_cfg(name)['enable_masking'] else None
    cots = [torch.from_numpy(a).to(device) for a in casegen.make_cotangents(name)]
    feats, states = m.forward_sequence(xs, None, masks)
    loss = sum((feats[s + 1].float() * cots[s]).sum() for s in range(4))
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert all(g is not None for g in grads.values())
    feats_all = [{s + 1: feats[s + 1][t] for s in range(4)} for t in range(c['T'])]
    out = summarize(name, feats_all, states, grads)
    if with_batch2:
        with torch.no_grad():
            first = torch.tensor([True] + [False] * (c['B'] - 1), device=device)
            st2 = []
            for h, cc in states:
                h2, c2 = h.detach().clone(), cc.detach().clone()
                h2[first] = 0           # same in-place reset as modules/utils/detection.py:96-113
                c2[first] = 0
                st2.append((h2, c2))
            feats2, _ = m.forward_sequence(xs, st2, masks)
        out.update(summarize(name, [{s + 1: feats2[s + 1][t] for s in range(4)} for t in range(c['T'])], None, None,
                             prefix='b2_'))
    return out


@pytest.mark.parametrize('name', EMU_CASES)
def test_backbone_fp32_vs_reference_golden_emu(name):
    """fp32 HIP kernel sources (CPU SIMT emulator) vs the unmodified reference's recorded outputs/gradients."""
    from rvt_amd import _lib
    from tests.backends import emu_library
    _lib._install_test_library(emu_library())
    try:
        got = run_hip_case(name, torch.device('cpu'), torch.float32)
    finally:
        _lib._install_test_library(None)
    compare(got, load_golden(name), rtol=1e-3, what=f'hip(emu) vs reference [{name}]', grad_rtol=1e-3)


@pytest.mark.parametrize('dtype,rtol,grtol,route', [(torch.float32, 1e-3, 1e-3, 1), (torch.bfloat16, 3e-2, 5e-2, 1),
                                                    (torch.bfloat16, 3e-2, 5e-2, -1)])
def test_backbone_fused_mlp_path_emu(dtype, rtol, grtol, route):
    """Fused-MLP routes against the reference golden: tuning.route_fused_mlp = 1 (every supported half; stages with C in {64,128})
    and the bf16 default -1 (C = 128: fused forward that saves only the pre-activation + op-by-op backward with GELU on load)."""
    from rvt_amd import _lib, tuning
    from tests.backends import emu_library
    _lib._install_test_library(emu_library())
    try:
        with tuning.override(route_fused_mlp=route):
            got = run_hip_case('micro', torch.device('cpu'), dtype, with_batch2=False)
    finally:
        _lib._install_test_library(None)
    compare(got, load_golden('micro'), rtol=rtol, what='fused-MLP route vs reference [micro]', grad_rtol=grtol)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,rtol,grtol', [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 3e-2, 5e-2)])
def test_backbone_fused_mlp_path_gpu(dtype, rtol, grtol):
    from rvt_amd import tuning
    with tuning.override(route_fused_mlp=1):
        got = run_hip_case('micro', torch.device('cuda', 0), dtype, with_batch2=False)
    compare(got, load_golden('micro'), rtol=rtol, what='fused-MLP route vs reference [micro]', grad_rtol=grtol)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ALL_CASES)
def test_backbone_fp32_vs_reference_golden_gpu(name):
    """north_star bar: feature maps + hidden states within 1e-3 rel of the reference CPU path in fp32
    (measured ~1e-5); gradients of every parameter to the same bar."""
    got = run_hip_case(name, torch.device('cuda', 0), torch.float32)
    worst = compare(got, load_golden(name), rtol=1e-3, what=f'hip vs reference [{name}]', grad_rtol=1e-3)
    print(f'{name}: worst err/tol = {worst:.3e}')


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['micro', 'tiny_gen1_gamma', 'base_qvga', 'base_1mpx', 'base_gen1'])
def test_backbone_bf16_vs_reference_golden_gpu(name):
    """bf16 performance mode against the fp32 reference: stated looser bound (bf16 has 8 mantissa bits;
    errors accumulate through 4 stages x T recurrent steps): 3e-2 of the tensor scale on features, 5e-2 on gradients
    (round 2, tightened from 4e-2 / 8e-2: the worst measured ratios on MI355X are 0.86 of 2.5e-2 / 4e-2, profiles/r2/bf16_worst.txt)."""
    got = run_hip_case(name, torch.device('cuda', 0), torch.bfloat16, with_batch2=False)
    compare(got, load_golden(name), rtol=3e-2, what=f'hip bf16 vs reference [{name}]', grad_rtol=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['base_1mpx_t21', 'tiny_gen1_t21'])
def test_backbone_bf16_vs_reference_golden_t21_gpu(name):
    """bf16 against the fp32 REFERENCE over the full recurrent depth of the BASELINE GPU configs (T = 21).  Bound: 4e-2 of the
    tensor scale on features / final cell states (the per-step drift of bf16 storage against fp32 saturates at ~2.1e-2 after a
    few steps, tests/test_drift.py; the reference comparison adds the fp32 path's own ~1e-5), 5e-2 on gradient norms / samples."""
    got = run_hip_case(name, torch.device('cuda', 0), torch.bfloat16, with_batch2=False)
    worst = compare(got, load_golden(name), rtol=4e-2, what=f'hip bf16 vs reference [{name}]', grad_rtol=5e-2)
    print(f'{name}: worst err/tol = {worst:.3e}')
