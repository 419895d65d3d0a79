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


# ---- training-side driver (round 6): rvt_stage_seq_train_fwd / rvt_stage_seq_bwd against the Python host loop -----------------------
def _train_run(name, dev, dtype, with_state):
    m = build_model(name, dev, dtype)
    xs = torch.from_numpy(casegen.make_inputs(name)).to(dev)
    cots = [torch.from_numpy(a).to(dev) for a in casegen.make_cotangents(name)]
    states = None
    if with_state:                       # incoming states with gradient flow into them (BPTT across calls within a batch)
        with torch.no_grad():
            _, st = m.forward_sequence(xs, None)
        states = [(h.detach().clone().requires_grad_(True), c.detach().clone().requires_grad_(True)) for h, c in st]
    feats, st_out = m.forward_sequence(xs, states)
    loss = sum((feats[s + 1].float() * cots[s]).sum() for s in range(4)) + sum(c.float().sum() * 0.5 for _, c in st_out)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    sgrads = [] if states is None else [t.grad.clone() for hc in states for t in hc]
    return feats, st_out, grads, sgrads


@pytest.mark.parametrize('name,dtype,with_state', [('micro', torch.float32, False), ('micro', torch.bfloat16, True), ('base_qvga', torch.bfloat16, False),
                                                   ('micro_dh24', torch.float32, True), ('tiny_gen1', torch.bfloat16, False)])
def test_train_stage_driver_matches_host_loop(backend, name, dtype, with_state):
    """The training forward + BPTT backward through the C-side stage driver (one library call per stage and direction) must be
    BIT-identical to the Python host loop over the same operators: features, final states, every parameter gradient, and the
    gradients of the incoming states."""
    dev = backend
    if dev.type == 'cpu' and name in ('base_qvga', 'tiny_gen1'):
        pytest.skip('minutes on the CPU emulator; runs on the GPU backend')
    outs = {}
    for drv in (0, 1):
        with tuning.override(route_stage_driver_train=drv):
            outs[drv] = _train_run(name, dev, dtype, with_state)
    a, b = outs[0], outs[1]
    for s in (1, 2, 3, 4):
        assert torch.equal(a[0][s], b[0][s]), f'features of stage {s}'
    for s in range(4):
        for x, y in zip(a[1][s], b[1][s]):
            assert torch.equal(x, y), f'final states of stage {s + 1}'
    for k in a[2]:
        if dev.type == 'cpu':
            assert torch.equal(a[2][k], b[2][k]), f'gradient of {k}: max diff {float((a[2][k] - b[2][k]).abs().max()):.3e}'
        else:       # on the GPU the LayerNorm / bias gradients fold through fp32 atomics: their order differs from run to run
            err = float((a[2][k] - b[2][k]).abs().max()) / max(float(a[2][k].abs().max()), 1e-30)
            assert err <= 2e-5, f'gradient of {k}: rel diff {err:.3e}'
    for i, (x, y) in enumerate(zip(a[3], b[3])):
        assert torch.equal(x, y), f'gradient of incoming state tensor {i}'


def test_train_stage_driver_is_taken_and_declines(backend):
    """The driver is the default training route where covered (StageSaved.train is set), and hands token masks / the DWS-ConvLSTM back
    to the host loop without an error."""
    from rvt_amd import stage as stage_mod
    dev = backend
    seen = []
    orig = stage_mod.stage_seq_forward

    def spy(*a, **k):
        out = orig(*a, **k)
        seen.append(out[2] is not None and out[2].train is not None)
        return out
    stage_mod.stage_seq_forward = spy
    import rvt_amd.backbone as bb
    old = bb.stage_seq_forward
    bb.stage_seq_forward = spy
    try:
        for name, want in (('micro', True), ('micro_mask', None), ('micro_dws_hidden', False)):
            seen.clear()
            m = build_model(name, dev, torch.float32)
            xs = torch.from_numpy(casegen.make_inputs(name)).to(dev)
            masks = torch.from_numpy(casegen.make_token_masks(name)).to(dev) if name == 'micro_mask' else None
            feats, _ = m.forward_sequence(xs, None, masks)
            sum(feats[s].float().sum() for s in (1, 2, 3, 4)).backward()
            assert all(p.grad is not None for p in m.parameters())
            if want is True:
                assert all(seen) and len(seen) == 4, seen
            elif want is False:
                assert not any(seen), seen
            else:
                assert seen[0] is False and all(seen[1:]), seen       # the mask only touches stage 1
    finally:
        stage_mod.stage_seq_forward = orig
        bb.stage_seq_forward = old
