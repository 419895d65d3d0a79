"""bench.py's algorithmic FLOP / byte model (opmodel.py) must understand the arguments of EVERY launch the backbone issues:
run a small training step on the emulator build with a recording hook on the C-ABI call and price each launch."""
import torch

import opmodel
from rvt_amd import _lib, tuning
from tests.backends import emu_library
from tests.test_backbone import run_hip_case

TABLES = {'rvt_pack_table', 'rvt_layerscale_grad_table', 'rvt_state_reset_masked', 'rvt_gather_frames'}


def _record(case, dtype, **route):
    calls = []
    orig = _lib.call

    def rec(name, *args):
        calls.append((name, args))
        return orig(name, *args)
    _lib._install_test_library(emu_library())
    _lib.call = rec
    try:
        with tuning.override(route_stage_driver_train=0, **route):       # (the C-side stage driver issues its launches inside one library call)
            run_hip_case(case, torch.device('cpu'), dtype, with_batch2=False)
    finally:
        _lib.call = orig
        _lib._install_test_library(None)
    return calls


def test_every_launch_of_a_training_step_is_priced():
    seen = set()
    for case, dtype, route in (('micro', torch.bfloat16, {}), ('micro', torch.float32, dict(route_lstm_scan=0, route_fused_mlp=0, route_attn_block=0)),
                               ('micro_dws_xh', torch.float32, {}), ('micro_mask', torch.float32, {})):
        for name, args in _record(case, dtype, **route):
            m = opmodel.model(name, args)
            if name in TABLES:
                assert m is None
                continue
            assert m is not None, f'{name}: no algorithmic model'
            fl, by = m
            assert fl >= 0 and by > 0, (name, fl, by)
            seen.add(name)
            if name == 'rvt_linear_fwd':
                M, N, K = args[5], args[6], args[7]
                assert fl == 2.0 * M * N * K
    assert {'rvt_linear_wgrad', 'rvt_lstm_scan_bwd', 'rvt_attn_bwd', 'rvt_conv_fwd', 'rvt_layernorm_bwd', 'rvt_lstm_fwd',
            'rvt_dwconv_fwd', 'rvt_token_mask_fwd'} <= seen, seen


def test_roofline_entry_picks_the_binding_roof():
    r = opmodel.roofline_entry('x', flops=1e12, bytes_=1e10, ms=1.0, launches=1, dtype='bf16')      # AI 100 < ridge 312.5
    assert r['bound'] == 'hbm' and abs(r['achieved'] - 1e4) < 1 and r['unit'] == 'GB/s'
    r = opmodel.roofline_entry('x', flops=1e13, bytes_=1e10, ms=10.0, launches=2, dtype='bf16')     # AI 1000
    assert r['bound'] == 'mfma' and abs(r['achieved'] - 1000.0) < 1 and r['frac'] == 0.4
