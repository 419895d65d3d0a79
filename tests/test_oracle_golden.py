"""Pin the CPU oracle (oracle/rvt_oracle.py) against fixtures recorded from the unmodified
reference (oracle/make_golden.py).  Runs on CPU; no GPU, no reference tree needed."""
import pytest
import torch

from tests import casegen
from tests.harness import compare, load_golden, oracle_run


@pytest.mark.parametrize('name', list(casegen.CASES))
def test_oracle_matches_reference_golden(name):
    torch.set_num_threads(max(1, torch.get_num_threads()))
    want = load_golden(name)
    # (T = 21 cases: the carried-state second batch is covered by the short cases; skipping it keeps the CPU suite in minutes)
    # (the 1 Mpx T = 21 case runs the oracle in its ATen-op mode — F.conv2d / F.layer_norm / F.gelu instead of the explicit
    # arithmetic, 60 s instead of 160 s; the explicit forms are pinned by the other cases and by the equivalence test below)
    got = oracle_run(name, torch.float32, with_batch2=not name.endswith('_t21'),
                     conv_impl='aten' if name == 'base_1mpx_t21' else 'im2col')
    # fp32 vs fp32, different op order (im2col einsum vs oneDNN conv, gather vs permute): 2e-4 is ample
    compare(got, want, rtol=2e-4, what=f'oracle vs reference [{name}]', grad_rtol=5e-4)


def test_oracle_fp64_agrees_with_fp32_reference():
    """fp64 oracle vs fp32 reference golden: bounds the reference's own fp32 round-off."""
    want = load_golden('micro')
    got = oracle_run('micro', torch.float64)
    compare(got, want, rtol=1e-4, what='fp64 oracle vs fp32 reference', grad_rtol=3e-4)


def test_layerscale_blind_spot_is_real():
    """with γ=1e-5 the attention/MLP path is invisible at 1e-3; with γ~U(0.5,1.5) it is not.
    Guards the test-suite itself: the γ-randomised cases must be the ones that gate parity."""
    a = load_golden('micro')
    b = load_golden('micro_default_gamma')
    assert abs(a['loss'] - b['loss']) > 1e-2


def test_oracle_aten_timing_mode_matches_reference_golden():
    """bench.py's cpu_baseline leg times the oracle with OracleCfg.conv_impl='aten' (F.conv2d / F.layer_norm / F.gelu, the ops the
    reference calls, instead of the explicit arithmetic): same numbers."""
    from oracle import rvt_oracle as O
    from tests import casegen as cg
    name = 'micro'
    c, cfgd = cg.CASES[name], cg.case_cfg(name)
    cfg = O.OracleCfg(**{k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in cfgd.items()}, conv_impl='aten')
    params = {k: torch.from_numpy(v) for k, v in cg.make_params(cfgd, seed=0, gamma=c['gamma']).items()}
    xs = torch.from_numpy(cg.make_inputs(name))
    try:
        fa, sa = O.sequence_forward(xs, None, params, cfg, c['in_res'])
        cfg.conv_impl = 'im2col'
        fb, sb = O.sequence_forward(xs, None, params, cfg, c['in_res'])
    finally:
        O._ATEN[0] = False
    for t in range(c['T']):
        for s in (1, 2, 3, 4):
            assert torch.allclose(fa[t][s], fb[t][s], rtol=1e-4, atol=1e-5), (t, s)
