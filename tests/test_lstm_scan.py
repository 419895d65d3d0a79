"""ConvLSTM scan kernels (time loop in the kernel, rvt_amd/csrc/lstm_scan.hpp) against (a) a plain fp64 restatement of
reference models/layers/rnn.py:52-67 differentiated by autograd and (b) the per-step kernels they replace."""
import pytest
import torch

from rvt_amd import ops, weights
from tests import bounds
from tests.backends import backend  # noqa: F401

TOL = {torch.float32: 3e-5, torch.bfloat16: 3e-2}


def rnd(shape, dev, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt).to(dev)


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-9))


def near(got, want, dt, what, mult=1.0):
    """fp32: 3e-5 * mult of the reference's max.  bf16: the bound measured for this check (tests/bounds.py) - no multiplier."""
    err = rel(got, want)
    if dt == torch.bfloat16:
        bounds.check(err, what)
    else:
        assert err <= TOL[dt] * mult, (what, err)


def ref_lstm(x, h0, c0, w, b):
    """x (T,M,C) fp64; rnn.py:52-67 with a 1x1 conv == a linear over [x|h]."""
    C = x.shape[-1]
    hs, cs = [], []
    h, c = h0, c0
    for t in range(x.shape[0]):
        z = torch.cat([x[t], h], -1) @ w.t() + b
        f, i, o = torch.sigmoid(z[:, :C]), torch.sigmoid(z[:, C:2 * C]), torch.sigmoid(z[:, 2 * C:3 * C])
        g = torch.tanh(z[:, 3 * C:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        hs.append(h)
        cs.append(c)
    return torch.stack(hs), torch.stack(cs)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,M,T', [(32, 200, 3), (64, 333, 4), (128, 70, 2), (64, 31, 2), (64, 33, 2), (64, 63, 2), (64, 65, 3), (64, 129, 2),
                                   (64, 257, 2), (128, 33, 2), (32, 65, 2)])
@pytest.mark.parametrize('zero_state', [False, True])
def test_lstm_scan_fwd_bwd(backend, dt, C, M, T, zero_state):
    assert ops.lstm_scan_supported(dt, C)
    dev = backend
    x = rnd((T, M, C), dev, dt, 1)
    h0 = torch.zeros(M, C, dtype=dt, device=dev) if zero_state else rnd((M, C), dev, dt, 2, 0.5)
    c0 = None if zero_state else rnd((M, C), dev, torch.float32, 3, 0.7)
    w = rnd((4 * C, 2 * C), dev, dt, 4, 1.0 / (2 * C) ** 0.5 * 2)
    b = rnd((4 * C,), dev, torch.float32, 5, 0.2)
    dH = rnd((T, M, C), dev, dt, 6)
    dc_last = rnd((M, C), dev, torch.float32, 7)

    Hall = torch.empty(T + 1, M, C, dtype=dt, device=dev)
    Hall[0].copy_(h0)
    c_last = torch.empty(M, C, dtype=torch.float32, device=dev)
    Csave = torch.empty(T, M, C, dtype=dt, device=dev)
    ops.lstm_scan_fwd(x, Hall, c0, c_last, Csave, w, b)

    xr, wr, br = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    h0r = h0.double().cpu().requires_grad_(True)
    c0r = (torch.zeros(M, C, dtype=torch.float64) if c0 is None else c0.double().cpu()).requires_grad_(True)
    hs, cs = ref_lstm(xr, h0r, c0r, wr, br)
    tol = TOL[dt]
    near(Hall[1:], hs, dt, 'h')
    near(c_last, cs[-1], dt, 'c_last')
    near(Csave, cs, dt, 'Csave')

    # inference flavour (no saved cell states) gives the same numbers
    Hall2 = torch.empty_like(Hall)
    Hall2[0].copy_(h0)
    c_last2 = torch.empty_like(c_last)
    ops.lstm_scan_fwd(x, Hall2, c0, c_last2, None, w, b)
    assert torch.equal(Hall2.cpu(), Hall.cpu()) and torch.equal(c_last2.cpu(), c_last.cpu())

    # the per-step kernels it replaces (gate-interleaved weights): same arithmetic, fp32 must agree to round-off
    perm = weights.lstm_gate_perm(C, dev)
    Hs = torch.empty_like(Hall)
    Hs[0].copy_(h0)
    Cs = torch.zeros(T + 1, M, C, dtype=torch.float32, device=dev)
    if c0 is not None:
        Cs[0].copy_(c0)
    for t in range(T):
        ops.lstm_fwd(x[t], Hs[t], Cs[t], w[perm].contiguous(), b[perm].contiguous(), Hs[t + 1], Cs[t + 1], None)
    if dt == torch.float32:
        assert rel(Hall[1:], Hs[1:]) <= 1e-6
    else:
        near(Hall[1:], Hs[1:], dt, 'h vs per-step kernels')

    # backward
    (hs * dH.double().cpu()).sum().backward(retain_graph=True)
    torch.autograd.backward([cs[-1]], [dc_last.double().cpu()])
    dx = torch.empty(T, M, C, dtype=dt, device=dev)
    dz = torch.empty(T, M, 4 * C, dtype=dt, device=dev)
    dh0 = torch.empty(M, C, dtype=dt, device=dev)
    dc0 = torch.empty(M, C, dtype=torch.float32, device=dev)
    ops.lstm_scan_bwd(x, Hall, Csave, c0, dH, dc_last, w, w.t().contiguous(), b, dx, dz, dh0, dc0)
    near(dx, xr.grad, dt, 'dx')
    near(dh0, h0r.grad, dt, 'dh0')
    near(dc0, c0r.grad, dt, 'dc0')
    # dz: check through the weight / bias gradients it produces (dW = sum_t dz_t^T [x_t | h_{t-1}])
    xh = torch.cat([x.double().cpu(), Hall[:T].double().cpu()], -1).reshape(T * M, 2 * C)
    dzc = dz.double().cpu().reshape(T * M, 4 * C)
    near(dzc.t() @ xh, wr.grad, dt, 'dW', mult=2.0)
    near(dzc.sum(0), br.grad, dt, 'db', mult=2.0)

    # saved-gates route (bf16, C = 128: weights in the register file): the forward also stores the activated gates, the reverse
    # scan reads them instead of recomputing (same numbers as the recompute route up to the bf16 rounding of the stored gates)
    if ops.lstm_scan_saves_gates(dt, C):
        gates = torch.empty(T, M, 4 * C, dtype=dt, device=dev)
        Hall3 = torch.empty_like(Hall)
        Hall3[0].copy_(h0)
        c_last3, Csave3 = torch.empty_like(c_last), torch.empty_like(Csave)
        ops.lstm_scan_fwd(x, Hall3, c0, c_last3, Csave3, w, b, gates_out=gates)
        assert torch.equal(Hall3.cpu(), Hall.cpu()) and torch.equal(Csave3.cpu(), Csave.cpu())
        dx3, dz3 = torch.empty_like(dx), torch.empty_like(dz)
        dh03, dc03 = torch.empty_like(dh0), torch.empty_like(dc0)
        ops.lstm_scan_bwd(x, Hall, Csave, c0, dH, dc_last, w, w.t().contiguous(), b, dx3, dz3, dh03, dc03, gates=gates)
        near(dx3, xr.grad, dt, 'gates dx')
        near(dh03, h0r.grad, dt, 'gates dh0')
        near(dc03, c0r.grad, dt, 'gates dc0')
        dzc3 = dz3.double().cpu().reshape(T * M, 4 * C)
        near(dzc3.t() @ xh, wr.grad, dt, 'gates dW', mult=2.0)
    else:
        assert dt == torch.float32 or C != 128

    # in-kernel weight gradients (bf16, LDS-resident weights): same dx / dh0 / dc0, dW and db accumulated (+=) without a dz tensor
    if ops.lstm_scan_wgrad_supported(dt, C, M):
        dw = torch.ones(4 * C, 2 * C, dtype=torch.float32, device=dev)
        db = torch.ones(4 * C, dtype=torch.float32, device=dev)
        dx2, dh02, dc02 = torch.empty_like(dx), torch.empty_like(dh0), torch.empty_like(dc0)
        ops.lstm_scan_bwd(x, Hall, Csave, c0, dH, dc_last, w, w.t().contiguous(), b, dx2, None, dh02, dc02, dw=dw, db=db)
        if C == 64:      # round 4: the in-kernel-gradient variant at C = 64 is the T-form kernel of lstm_scan2.hpp (other summation order)
            near(dx2, xr.grad, dt, 'v2 dx')
            near(dh02, h0r.grad, dt, 'v2 dh0')
            near(dc02, c0r.grad, dt, 'v2 dc0')
            near(dx2, dx, dt, 'v2 dx vs lstm_scan.hpp')
            near(dh02, dh0, dt, 'v2 dh0 vs lstm_scan.hpp')
            near(dc02, dc0, dt, 'v2 dc0 vs lstm_scan.hpp')
        else:
            assert torch.equal(dx2.cpu(), dx.cpu()) and torch.equal(dh02.cpu(), dh0.cpu()) and torch.equal(dc02.cpu(), dc0.cpu())
        near(dw - 1.0, wr.grad, dt, 'in-kernel dW', mult=2.0)
        near(db - 1.0, br.grad, dt, 'in-kernel db', mult=2.0)
    else:
        assert dt == torch.float32 or C > 64

    # no upstream cotangents at all -> every gradient is zero / finite
    ops.lstm_scan_bwd(x, Hall, Csave, c0, None, None, w, w.t().contiguous(), b, dx, dz, dh0, dc0)
    assert float(dx.float().abs().max()) == 0.0 and float(dz.float().abs().max()) == 0.0


# ---- wide stages (bf16, C = 256): weights streamed in operand order, gates saved for the reverse scan (csrc/lstm_scan3.hpp) ----
def _scan3_run(dev, C, M, T, zero_state, rb):
    from rvt_amd import tuning as tn
    dt = torch.bfloat16
    x = rnd((T, M, C), dev, dt, 1)
    h0 = torch.zeros(M, C, dtype=dt, device=dev) if zero_state else rnd((M, C), dev, dt, 2, 0.5)
    c0 = None if zero_state else rnd((M, C), dev, torch.float32, 3, 0.7)
    w = rnd((4 * C, 2 * C), dev, dt, 4, 1.0 / (2 * C) ** 0.5 * 2)
    b = rnd((4 * C,), dev, torch.float32, 5, 0.2)
    dH = None if zero_state else rnd((T, M, C), dev, dt, 6)
    dc_last = None if zero_state else rnd((M, C), dev, torch.float32, 7)
    with tn.override(lstm_scan3_rb256=rb, lstm_scan3_rb128=rb):
        assert ops.lstm_scan3_supported(dt, C)
        rows = ops.lstm_scan3_rows(C, M)
        assert rows >= M and rows % (32 * rb) == 0
        wp, wtp = ops.lstm_scan3_pack(w)
        Hall = torch.empty(T + 1, M, C, dtype=dt, device=dev)
        Hall[0].copy_(h0)
        c_last = torch.empty(M, C, dtype=torch.float32, device=dev)
        Csave = torch.empty(T, rows, C, dtype=dt, device=dev)
        gsave = torch.empty(T, rows, 4 * C, dtype=dt, device=dev)
        ops.lstm_scan3_fwd(x, Hall, c0, c_last, Csave, wp, b, gsave)
        # no-grad flavour: same numbers, nothing saved
        Hall2 = torch.empty_like(Hall)
        Hall2[0].copy_(h0)
        c_last2 = torch.empty_like(c_last)
        ops.lstm_scan3_fwd(x, Hall2, c0, c_last2, None, wp, b, None)
        assert torch.equal(Hall2.cpu(), Hall.cpu()) and torch.equal(c_last2.cpu(), c_last.cpu())
        dx = torch.empty(T, M, C, dtype=dt, device=dev)
        dz = torch.empty(T, M, 4 * C, dtype=dt, device=dev)
        dh0 = torch.empty(M, C, dtype=dt, device=dev)
        dc0 = torch.empty(M, C, dtype=torch.float32, device=dev)
        ops.lstm_scan3_bwd(gsave, Csave, c0, dH, dc_last, wtp, dx, dz, dh0, dc0)
    return dict(x=x, h0=h0, c0=c0, w=w, b=b, dH=dH, dc_last=dc_last, Hall=Hall, c_last=c_last, dx=dx, dz=dz, dh0=dh0, dc0=dc0)


@pytest.mark.parametrize('C', [256, 128])
@pytest.mark.parametrize('M,T,rb', [(200, 3, 2), (97, 4, 1), (64, 2, 2), (31, 2, 1), (333, 3, 2), (33, 2, 2), (63, 2, 1), (65, 2, 1), (129, 2, 2), (257, 2, 1)])
@pytest.mark.parametrize('zero_state', [False, True])
def test_lstm_scan3_fwd_bwd(backend, C, M, T, rb, zero_state):
    dt = torch.bfloat16
    r = _scan3_run(backend, C, M, T, zero_state, rb)
    x, h0, c0, w, b = r['x'], r['h0'], r['c0'], r['w'], r['b']
    xr, wr, br = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    h0r = h0.double().cpu().requires_grad_(True)
    c0r = (torch.zeros(M, C, dtype=torch.float64) if c0 is None else c0.double().cpu()).requires_grad_(True)
    hs, cs = ref_lstm(xr, h0r, c0r, wr, br)
    tol = TOL[dt]
    near(r['Hall'][1:], hs, dt, 'h')
    near(r['c_last'], cs[-1], dt, 'c_last')
    if zero_state:         # no upstream cotangents at all -> every gradient is exactly zero
        for k in ('dx', 'dz', 'dh0', 'dc0'):
            assert float(r[k].float().abs().max()) == 0.0, k
        return
    (hs * r['dH'].double().cpu()).sum().backward(retain_graph=True)
    torch.autograd.backward([cs[-1]], [r['dc_last'].double().cpu()])
    near(r['dx'], xr.grad, dt, 'dx')
    near(r['dh0'], h0r.grad, dt, 'dh0')
    near(r['dc0'], c0r.grad, dt, 'dc0')
    xh = torch.cat([x.double().cpu(), r['Hall'][:T].double().cpu()], -1).reshape(T * M, 2 * C)
    dzc = r['dz'].double().cpu().reshape(T * M, 4 * C)
    near(dzc.t() @ xh, wr.grad, dt, 'dW')
    near(dzc.sum(0), br.grad, dt, 'db')


def test_lstm_scan3_matches_per_step_kernels(backend):
    """Same arithmetic as the per-step kernels it replaces (gate-interleaved GEMM + gate kernels): bf16 round-off apart."""
    C, M, T, dt, dev = 256, 130, 3, torch.bfloat16, backend
    r = _scan3_run(dev, C, M, T, False, 2)
    perm = weights.lstm_gate_perm(C, dev)
    Hs = torch.empty_like(r['Hall'])
    Hs[0].copy_(r['h0'])
    Cs = torch.zeros(T + 1, M, C, dtype=torch.float32, device=dev)
    Cs[0].copy_(r['c0'])
    for t in range(T):
        ops.lstm_fwd(r['x'][t], Hs[t], Cs[t], r['w'][perm].contiguous(), r['b'][perm].contiguous(), Hs[t + 1], Cs[t + 1], None)
    near(r['Hall'][1:], Hs[1:], dt, 'h vs per-step kernels')
    near(r['c_last'], Cs[T], dt, 'c_last vs per-step kernels')
