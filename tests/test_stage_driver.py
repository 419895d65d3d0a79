"""The C-side stage driver (include/rvt_hip.h: rvt_stage_seq_fwd, csrc/capi_stage.hip) against the Python host loop of
rvt_amd/stage.py: the no-grad forward launches the same operators in the same order either way, so features and states must be
BIT-identical; and against the reference golden through the normal parity path (the driver is the default no-grad route)."""
import pytest
import torch

from rvt_amd import tuning
from tests import casegen
from tests.backends import backend  # noqa: F401
from tests.test_backbone import build_model


@pytest.mark.parametrize('name,dtype', [('micro', torch.float32), ('micro', torch.bfloat16), ('base_qvga', torch.bfloat16),
                                        ('micro_dh24', torch.float32)])
def test_stage_driver_matches_host_loop(backend, name, dtype):
    dev = backend
    m = build_model(name, dev, dtype).eval()
    xs = torch.from_numpy(casegen.make_inputs(name)).to(dev)
    T = xs.shape[0]
    outs = {}
    for drv in (0, 1):
        with tuning.override(route_stage_driver=drv), torch.no_grad():
            feats_seq, st_seq = m.forward_sequence(xs, None)                  # whole sequence at once, zero initial state
            states = None
            per_step = []
            for t in range(T):                                               # streaming: T = 1 calls with carried states
                f, states = m(xs[t], states)
                per_step.append(f)
            outs[drv] = (feats_seq, st_seq, per_step, states)
    a, b = outs[0], outs[1]
    for s in (1, 2, 3, 4):
        assert torch.equal(a[0][s], b[0][s]), f'sequence features of stage {s} differ between driver and host loop'
        for t in range(T):
            assert torch.equal(a[2][t][s], b[2][t][s]), f'streaming features, step {t} stage {s}'
    for s in range(4):
        for x, y in zip(a[1][s] + a[3][s], b[1][s] + b[3][s]):
            assert torch.equal(x, y), f'states of stage {s + 1}'
    # streaming == sequence (same arithmetic; fp32 tight, bf16 the scan / per-step kernels differ in summation order)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for s in (1, 2, 3, 4):
        for t in range(T):
            want = b[0][s][t].float()
            err = (b[2][t][s].float() - want).abs().max().item() / max(want.abs().max().item(), 1e-9)
            assert err <= tol, (s, t, err)


def test_stage_driver_declines_what_it_does_not_cover(backend):
    """Token masks (and the DWS-ConvLSTM) stay on the operator-by-operator host loop: same results as before, no error."""
    dev = backend
    m = build_model('micro_mask', dev, torch.float32).eval()
    xs = torch.from_numpy(casegen.make_inputs('micro_mask')).to(dev)
    masks = torch.from_numpy(casegen.make_token_masks('micro_mask')).to(dev)
    with torch.no_grad():
        f1, _ = m.forward_sequence(xs, None, masks)
        with tuning.override(route_stage_driver=0):
            f0, _ = m.forward_sequence(xs, None, masks)
    for s in (1, 2, 3, 4):
        assert torch.equal(f0[s], f1[s])
    m2 = build_model('micro_dws_hidden', dev, torch.float32).eval()
    x2 = torch.from_numpy(casegen.make_inputs('micro_dws_hidden')).to(dev)
    with torch.no_grad():
        g1, _ = m2.forward_sequence(x2, None)
        with tuning.override(route_stage_driver=0):
            g0, _ = m2.forward_sequence(x2, None)
    for s in (1, 2, 3, 4):
        assert torch.equal(g0[s], g1[s])
