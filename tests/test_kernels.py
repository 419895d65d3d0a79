"""Per-kernel parity: every C-ABI entry point against a plain fp32/fp64 torch restatement of the op,
on the CPU emulator build (always) and on the real gfx950 build (-m gpu)."""
import math

import pytest
import torch
import torch.nn.functional as F

from rvt_amd import ops, tuning, weights
from tests import bounds
from tests.backends import backend  # noqa: F401

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def rnd(shape, dev, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt).to(dev)


def close(got, want, dt, what, scale=None, f32_mult=1.0):
    """fp32: |got - want|_max <= 2e-5 * f32_mult of the reference scale.  bf16: the bound MEASURED for this very check (1.5 x the
    error seen on the backend, tests/bounds.py; 2e-2 without an entry) - no multipliers."""
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    s = scale if scale is not None else max(want.abs().max().item(), 1e-6)
    err = (got - want).abs().max().item() / s
    if dt == torch.bfloat16:
        bounds.check(err, what)
    else:
        assert err <= TOL[dt] * f32_mult, f'{what}: rel err {err:.3e} (tol {TOL[dt] * f32_mult:.1e})'


def f64(t):
    return t.detach().double().cpu()


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('u8', [True, False])
@pytest.mark.parametrize('shape', [((3, 20, 5, 7), (8, 8)), ((2, 20, 6, 264), (8, 272)), ((1, 3, 4, 130), (4, 136))])
def test_prepack(backend, dt, u8, shape):
    (F, Cin, h, w), (H, W) = shape
    g = torch.Generator().manual_seed(0)
    src = torch.randint(0, 11, (F, Cin, h, w), generator=g, dtype=torch.uint8)
    if not u8:
        src = src.float()
    out = ops.prepack_input(src.to(backend), H, W, 24, dt)
    want = torch.zeros(F, H, W, 24)
    want[:, :h, :w, :Cin] = src.float().permute(0, 2, 3, 1)
    close(out, want, dt, 'prepack', f32_mult=0.0)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('M,N,K,gelu', [(200, 72, 40, False), (130, 136, 64, True), (64, 16, 16, False), (257, 264, 136, False),
                                        (2100, 520, 72, False),      # 17 x 5 tiles on the 8-workgroup test grid: XCD-grouped panel walk (gemm.hpp, mode 3)
                                        (1600, 264, 40, False), (1000, 40, 72, True),
                                        (300, 512, 136, False)])     # N a multiple of 256: the 128 x 256 tile where it is enabled
def test_linear_fwd(backend, dt, M, N, K, gelu):
    x, w = rnd((M, K), backend, dt, 1), rnd((N, K), backend, dt, 2, 0.3)
    b = rnd((N,), backend, torch.float32, 3)
    y = ops.linear_fwd(x, w, b, gelu_in=gelu)
    xa = f64(x)
    if gelu:
        xa = F.gelu(xa)
        if dt == torch.bfloat16:
            xa = xa.to(dt).double()
    close(y, xa @ f64(w).t() + f64(b), dt, 'linear_fwd')


PP_CASES = [(256, 256, 64), (700, 512, 192), (2500, 256, 128), (1300, 768, 256), (257, 256, 64), (289, 512, 128)]


@pytest.mark.parametrize('M,N,K', PP_CASES)
def test_ppgemm_routes(backend, M, N, K):
    """The 256 x 256 LDS-DMA GEMM (csrc/ppgemm.hpp) behind the linear entry points (bf16, N % 256 == 0, K % 64 == 0): every
    epilogue it is wired to, ragged last row panel, several tiles per workgroup, against fp64."""
    dt = torch.bfloat16
    x, w = rnd((M, K), backend, dt, 1), rnd((N, K), backend, dt, 2, 0.3)
    b, gam = rnd((N,), backend, torch.float32, 3), rnd((N,), backend, torch.float32, 4)
    res = rnd((M, N), backend, dt, 5)
    ref = f64(x) @ f64(w).t()
    close(ops.linear_fwd(x, w, b), ref + f64(b), dt, 'ppgemm linear_fwd')
    close(ops.linear_fwd(x, w, None), ref, dt, 'ppgemm linear_fwd (no bias)')
    close(ops.linear_scale_res_fwd(x, w, b, gam, res), f64(res) + f64(gam) * (ref + f64(b)), dt, 'ppgemm linear_scale_res_fwd')
    g, gp = ops.linear_gelu_fwd(x, w, b, want_grad=True)
    pre = (ref + f64(b)).requires_grad_(True)
    gr = F.gelu(pre)
    gr.sum().backward()
    close(g, gr, dt, 'ppgemm linear_gelu_fwd g')
    close(gp, pre.grad, dt, 'ppgemm linear_gelu_fwd gp')
    # input gradients: dx[M][N] = dy[M][K'] . wt[N][K']^T  (the entry point's (N, K) are this product's (K', N))
    close(ops.linear_dgrad(x, w), ref, dt, 'ppgemm linear_dgrad')
    close(ops.linear_dgrad(x, w, add=res), ref + f64(res), dt, 'ppgemm linear_dgrad + add')
    close(ops.linear_dgrad(x, w, mul=res), ref * f64(res), dt, 'ppgemm linear_dgrad * mul')
    p = f64(res).requires_grad_(True)
    F.gelu(p).sum().backward()
    close(ops.linear_dgrad(x, w, res), ref * p.grad, dt, "ppgemm linear_dgrad * gelu'")


@pytest.mark.parametrize('M,N,K', [(1000, 256, 256), (700, 512, 256), (1500, 256, 768), (600, 1024, 512)])
def test_ppgemm_tn_routes(backend, M, N, K):
    """Weight-gradient variant of the LDS-DMA GEMM (csrc/ppgemm_tn.hpp) behind rvt_linear_wgrad / rvt_lstm_wgrad (bf16, output
    dims multiples of 256): ragged last token slice, += semantics, bias column sums, the [x | h] two-matrix operand."""
    dt = torch.bfloat16
    dy, x = rnd((M, N), backend, dt, 1), rnd((M, K), backend, dt, 2)
    dw = torch.full((N, K), 0.25, device=backend)
    cs = torch.full((N,), -1.0, device=backend)
    ops.linear_wgrad(dy, x, dw, colsum_out=cs)
    close(dw - 0.25, f64(dy).t() @ f64(x), dt, 'ppgemm_tn dW')
    close(cs + 1.0, f64(dy).sum(0), dt, 'ppgemm_tn colsum', f32_mult=0.2)
    dw2 = torch.zeros(N, K, device=backend)
    ops.linear_wgrad(dy, x, dw2)
    close(dw2, f64(dy).t() @ f64(x), dt, 'ppgemm_tn dW (no colsum)')
    if N == 4 * (K // 2) and (K // 2) % 64 == 0:         # ConvLSTM shape: dz [M][4C], [x | h] two [M][C] operands (C = 128: the cut falls inside the one k tile)
        C = K // 2
        xs, hs = x[:, :C].contiguous(), x[:, C:].contiguous()
        dw3, db3 = torch.zeros(N, K, device=backend), torch.zeros(N, device=backend)
        ops.lstm_wgrad(dy, xs, hs, dw3, db3)
        close(dw3, f64(dy).t() @ f64(x), dt, 'ppgemm_tn lstm dW')
        close(db3, f64(dy).sum(0), dt, 'ppgemm_tn lstm colsum', f32_mult=0.2)


@pytest.mark.parametrize('dt', DTYPES)
def test_linear_gelu_and_mul(backend, dt):
    M, N, K = 150, 136, 40
    x, w, b = rnd((M, K), backend, dt, 1), rnd((N, K), backend, dt, 2, 0.4), rnd((N,), backend, torch.float32, 3)
    g, gp = ops.linear_gelu_fwd(x, w, b, want_grad=True)
    pre = (f64(x) @ f64(w).t() + f64(b)).requires_grad_(True)
    gr = F.gelu(pre)
    gr.sum().backward()
    close(g, gr, dt, 'linear_gelu_fwd g')
    close(gp, pre.grad, dt, 'linear_gelu_fwd gp')
    g2, none = ops.linear_gelu_fwd(x, w, b, want_grad=False)
    assert none is None and torch.equal(g2.cpu(), g.cpu())
    dy, wt = rnd((M, 72), backend, dt, 4), rnd((N, 72), backend, dt, 5, 0.2)
    dx = ops.linear_dgrad(dy, wt, mul=gp)
    close(dx, (f64(dy) @ f64(wt).t()) * f64(gp), dt, 'linear_dgrad mul')


@pytest.mark.parametrize('dt,C', [(torch.float32, 64), (torch.bfloat16, 64), (torch.bfloat16, 128)])
@pytest.mark.parametrize('M', [300, 1000, 31, 33, 63, 65, 129, 257])
def test_mlp_fused(backend, dt, C, M):
    """Fused MLP forward (+ saved GELU/GELU') and fused backward dgrad chain vs fp64 autograd and vs the op-by-op chain."""
    assert ops.mlp_fused_supported(dt, C)
    x = rnd((M, C), backend, dt, 1, 1.5)
    lw, lb = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0, rnd((C,), backend, torch.float32, 3, 0.2)
    w1, b1 = rnd((4 * C, C), backend, dt, 4, 0.2), rnd((4 * C,), backend, torch.float32, 5, 0.2)
    w2, b2 = rnd((C, 4 * C), backend, dt, 6, 0.1), rnd((C,), backend, torch.float32, 7, 0.2)
    gam = rnd((C,), backend, torch.float32, 8)
    dy = rnd((M, C), backend, dt, 9)
    y, g, gp, v2_saved = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=True, want_v2=True)
    y_inf, g_none, _ = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=False)
    assert g_none is None          # (C = 64: the nothing-saved flavour is the register-chained kernel of csrc/mlp_chain.hpp)

    xr = f64(x).requires_grad_(True)
    lwr, lbr = f64(lw).requires_grad_(True), f64(lb).requires_grad_(True)
    v2 = F.layer_norm(xr, (C,), lwr, lbr, 1e-5)
    pre = v2 @ f64(w1).t() + f64(b1)
    pre.retain_grad()
    h = F.gelu(pre)
    want = xr + f64(gam) * (h @ f64(w2).t() + f64(b2))
    want.backward(f64(dy))
    close(y, want, dt, 'mlp_fwd fused')
    close(y_inf, want, dt, 'mlp_fwd fused, nothing saved')
    close(v2_saved, v2, dt, 'mlp_fwd saved LayerNorm output')
    close(g, h, dt, 'mlp_fwd g')
    hp = f64(pre.detach()).requires_grad_(True)
    F.gelu(hp).sum().backward()
    close(gp, hp.grad, dt, 'mlp_fwd gp')
    # against the op-by-op HIP chain it replaces
    v2h = ops.layernorm_fwd(x, lw, lb, 1e-5)
    g2, gp2 = ops.linear_gelu_fwd(v2h, w1, b1, want_grad=True)
    y2 = ops.linear_scale_res_fwd(g2, w2, b2, gam, x)
    close(y, y2.double(), dt, 'mlp_fwd fused vs chain')
    # round 4: the pre-activation-only flavour (what the C = 128 training forward keeps) and the backward that consumes it:
    # GELU on load in the fc2 weight gradient, GELU' in the epilogue of the fc2 input gradient
    y_pre, hpre, none_gp, _ = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_pre=True, want_v2=True)
    assert none_gp is None
    close(y_pre, want, dt, 'mlp_fwd fused (pre-activation saved)')
    close(hpre, pre.detach(), dt, 'mlp_fwd saved pre-activation')
    s2 = torch.zeros(C, 4 * C, device=backend)
    ops.linear_wgrad(dy, hpre, s2, gelu_in=True)
    close(s2, f64(dy).t() @ h.detach(), dt, 'fc2 weight gradient from the pre-activation', f32_mult=2.0)
    dh_pre = ops.linear_dgrad(dy, (f64(w2) * f64(gam)[:, None]).t().to(dt).contiguous().to(backend), gelu_pre=hpre)
    close(dh_pre, pre.grad, dt, 'fc2 input gradient * GELU\'(pre-activation)', f32_mult=2.0)

    # backward dgrad chain: dh (grad of the pre-activation), dxmid, LayerNorm parameter grads
    w2g_t = (f64(w2) * f64(gam)[:, None]).t().to(dt).contiguous().to(backend)
    w1_t = f64(w1).t().to(dt).contiguous().to(backend)
    dlw, dlb = torch.zeros(C, device=backend), torch.zeros(C, device=backend)
    dh, dxm = ops.mlp_bwd_dgrad(dy, gp, x, lw, w2g_t, w1_t, dlw, dlb, 1e-5)
    close(dh, pre.grad, dt, 'mlp_bwd dh', f32_mult=2.0)
    close(dxm, xr.grad, dt, 'mlp_bwd dxmid', f32_mult=2.0)
    close(dlw, lwr.grad, dt, 'mlp_bwd dln_w', f32_mult=2.0)
    close(dlb, lbr.grad, dt, 'mlp_bwd dln_b', f32_mult=2.0)


@pytest.mark.parametrize('dt', DTYPES)
def test_linear_scale_res(backend, dt):
    M, N, K = 150, 48, 192
    x, w, res = rnd((M, K), backend, dt, 1), rnd((N, K), backend, dt, 2, 0.2), rnd((M, N), backend, dt, 5)
    b, gam = rnd((N,), backend, torch.float32, 3), rnd((N,), backend, torch.float32, 4)
    y = ops.linear_scale_res_fwd(x, w, b, gam, res, gelu_in=True)
    xa = F.gelu(f64(x))
    if dt == torch.bfloat16:
        xa = xa.to(dt).double()
    close(y, f64(res) + f64(gam) * (xa @ f64(w).t() + f64(b)), dt, 'linear_scale_res')


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('with_gelu', [False, True])
def test_linear_dgrad(backend, dt, with_gelu):
    M, N, K = 140, 96, 72
    dy, wt = rnd((M, N), backend, dt, 1), rnd((K, N), backend, dt, 2, 0.2)
    pre = rnd((M, K), backend, dt, 3) if with_gelu else None
    dx = ops.linear_dgrad(dy, wt, pre)
    want = f64(dy) @ f64(wt).t()
    if with_gelu:
        p = f64(pre).requires_grad_(True)
        F.gelu(p).sum().backward()
        want = want * p.grad
    close(dx, want, dt, 'linear_dgrad')


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('M,N,K,gelu', [(300, 24, 40, False), (1100, 136, 72, True), (70, 264, 16, False),
                                        (8200, 136, 72, False)])     # >= 16 K slices: XCD-dealt tile order
def test_linear_wgrad(backend, dt, M, N, K, gelu):
    dy, x = rnd((M, N), backend, dt, 1), rnd((M, K), backend, dt, 2)
    dw = torch.zeros(N, K, device=backend)
    cs = torch.zeros(N, device=backend)
    ops.linear_wgrad(dy, x, dw, gelu_in=gelu, colsum_out=cs)
    close(cs, f64(dy).sum(0), dt, 'linear_wgrad fused column sum', f32_mult=1.0)
    xa = f64(x)
    if gelu:
        xa = F.gelu(xa)
        if dt == torch.bfloat16:
            xa = xa.to(dt).double()
    close(dw, f64(dy).t() @ xa, dt, 'linear_wgrad')
    ops.linear_wgrad(dy, x, dw, gelu_in=gelu)          # accumulates
    close(dw, 2 * (f64(dy).t() @ xa), dt, 'linear_wgrad accumulate')


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('rows,C', [(37, 48), (100, 64), (9, 16), (21, 512), (33, 384)])
def test_layernorm(backend, dt, rows, C):
    x = rnd((rows, C), backend, dt, 1, 2.0)
    w, b = rnd((C,), backend, torch.float32, 2), rnd((C,), backend, torch.float32, 3)
    dy, dres = rnd((rows, C), backend, dt, 4), rnd((rows, C), backend, dt, 5)
    y = ops.layernorm_fwd(x, w, b, 1e-5)
    xr = f64(x).requires_grad_(True)
    wr, br = f64(w).requires_grad_(True), f64(b).requires_grad_(True)
    yr = F.layer_norm(xr, (C,), wr, br, 1e-5)
    close(y, yr, dt, 'ln_fwd')
    yr.backward(f64(dy))
    dw, db = torch.zeros(C, device=backend), torch.zeros(C, device=backend)
    dx = ops.layernorm_bwd(x, w, dy, dres, dw, db, 1e-5)
    close(dx, xr.grad + f64(dres), dt, 'ln_bwd dx')
    close(dw, wr.grad, dt, 'ln_bwd dw')
    close(db, br.grad, dt, 'ln_bwd db')


@pytest.mark.parametrize('C,K', [(64, 192), (64, 256), (128, 384), (128, 512)])
@pytest.mark.parametrize('M', [45, 300, 1000, 31, 33, 63, 65, 129, 257])
def test_linear_dgrad_ln(backend, C, K, M):
    """dx = dres + LN'(dy W; x) in one launch (csrc/dgrad_ln.hpp) vs fp64 autograd and vs the two-launch chain it replaces."""
    dt = torch.bfloat16
    assert ops.linear_dgrad_ln_supported(dt, C, K) and not ops.linear_dgrad_ln_supported(torch.float32, C, K)
    x = rnd((M, C), backend, dt, 1, 2.0)
    lw = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0
    w = rnd((K, C), backend, dt, 3, 0.2)
    dy, dres = rnd((M, K), backend, dt, 4), rnd((M, C), backend, dt, 5)
    dw, db = torch.zeros(C, device=backend), torch.zeros(C, device=backend)
    dx = ops.linear_dgrad_ln(dy, w, x, dres, lw, dw, db, 1e-5)
    xr, lwr, lbr = f64(x).requires_grad_(True), f64(lw).requires_grad_(True), torch.zeros(C, dtype=torch.float64, requires_grad=True)
    u = F.layer_norm(xr, (C,), lwr, lbr, 1e-5)
    (u @ f64(w).t()).backward(f64(dy))
    close(dx, xr.grad + f64(dres), dt, 'dgrad_ln dx')
    close(dw, lwr.grad, dt, 'dgrad_ln dln_w')
    close(db, lbr.grad, dt, 'dgrad_ln dln_b')
    # the chain: du (rounded to bf16) then the LayerNorm backward
    dw2, db2 = torch.zeros(C, device=backend), torch.zeros(C, device=backend)
    du = ops.linear_dgrad(dy, w.t().contiguous())
    dx2 = ops.layernorm_bwd(x, lw, du, dres, dw2, db2, 1e-5)
    close(dx, dx2.double(), dt, 'dgrad_ln vs chain dx')
    close(dw, dw2.double(), dt, 'dgrad_ln vs chain dln_w')
    # accumulation into existing parameter gradients, no residual
    dx3 = ops.linear_dgrad_ln(dy, w, x, None, lw, dw, db, 1e-5)
    close(dw, 2 * lwr.grad, dt, 'dgrad_ln dln_w accumulates')
    close(dx3, xr.grad, dt, 'dgrad_ln dx, no residual')


@pytest.mark.parametrize('C,K', [(64, 192), (128, 384)])
@pytest.mark.parametrize('M', [45, 300, 1000, 33, 129])
def test_linear_dgrad_preln(backend, C, K, M):
    """dy0 = LN'(dy W + add; y0) in one launch (csrc/dgrad_ln.hpp, INSIDE: the first block of a stage on the op-by-op attention
    route, carried through the down-sampling norm) vs fp64 autograd and vs the two launches it replaces."""
    dt = torch.bfloat16
    y0 = rnd((M, C), backend, dt, 1, 2.0)
    lw = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0
    w = rnd((K, C), backend, dt, 3, 0.2)
    dy, add = rnd((M, K), backend, dt, 4), rnd((M, C), backend, dt, 5)
    dw, db = torch.zeros(C, device=backend), torch.zeros(C, device=backend)
    out = ops.linear_dgrad_preln(dy, w, y0, add, lw, dw, db, 1e-5)
    yr, lwr, lbr = f64(y0).requires_grad_(True), f64(lw).requires_grad_(True), torch.zeros(C, dtype=torch.float64, requires_grad=True)
    u = F.layer_norm(yr, (C,), lwr, lbr, 1e-5)
    ((u @ f64(w).t() * f64(dy)).sum() + (u * f64(add)).sum()).backward()
    close(out, yr.grad, dt, 'dgrad_preln dy0')
    close(dw, lwr.grad, dt, 'dgrad_preln dln_w')
    close(db, lbr.grad, dt, 'dgrad_preln dln_b')
    dw2, db2 = torch.zeros(C, device=backend), torch.zeros(C, device=backend)
    dx = ops.linear_dgrad(dy, w.t().contiguous(), add=add)
    out2 = ops.layernorm_bwd(y0, lw, dx, None, dw2, db2, 1e-5)
    close(out, out2.double(), dt, 'dgrad_preln vs chain dy0')
    close(dw, dw2.double(), dt, 'dgrad_preln vs chain dln_w')
    close(db, db2.double(), dt, 'dgrad_preln vs chain dln_b')


def ref_attention(qkv, Fr, H, W, C, dh, ph, pw, window):
    """plain restatement of maxvit.py:273-304,343-354 on (F,H,W,3C) -> (F,H,W,C)"""
    heads = C // dh
    x = qkv.reshape(Fr, H, W, 3 * C)
    if window:
        t = x.reshape(Fr, H // ph, ph, W // pw, pw, 3 * C).permute(0, 1, 3, 2, 4, 5)
    else:
        t = x.reshape(Fr, ph, H // ph, pw, W // pw, 3 * C).permute(0, 2, 4, 1, 3, 5)
    t = t.reshape(-1, ph * pw, heads, 3, dh)
    q, k, v = t[:, :, :, 0].transpose(1, 2), t[:, :, :, 1].transpose(1, 2), t[:, :, :, 2].transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(-1, ph, pw, C)
    if window:
        o = o.reshape(Fr, H // ph, W // pw, ph, pw, C).permute(0, 1, 3, 2, 4, 5)
    else:
        o = o.reshape(Fr, H // ph, W // pw, ph, pw, C).permute(0, 3, 1, 4, 2, 5)
    return o.reshape(Fr, H, W, C)


ATTN_CASES = [  # F, H, W, C, dh, ph, pw
    (2, 4, 6, 32, 16, 2, 3),       # L=6   (NB=1), 2 heads
    (1, 12, 20, 64, 32, 6, 10),    # L=60  (NB=2), 1Mpx partition
    (1, 8, 20, 32, 32, 8, 10),     # L=80  (NB=3), Gen1 partition
    (1, 4, 6, 48, 24, 2, 3),       # dim_head 24 (RVT-Small)
    (1, 6, 10, 16, 8, 6, 10),      # dim_head 8, one partition per frame
]


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('window', [True, False])
@pytest.mark.parametrize('case', ATTN_CASES)
def test_attention(backend, dt, window, case):
    Fr, H, W, C, dh, ph, pw = case
    qkv = rnd((Fr, H, W, 3 * C), backend, dt, 1)
    dout = rnd((Fr, H, W, C), backend, dt, 2)
    out = ops.attn_fwd(qkv, Fr, H, W, C, dh, ph, pw, window)
    qr = f64(qkv).requires_grad_(True)
    want = ref_attention(qr, Fr, H, W, C, dh, ph, pw, window)
    close(out, want, dt, 'attn_fwd')
    want.backward(f64(dout))
    dq = ops.attn_bwd(qkv, dout, Fr, H, W, C, dh, ph, pw, window)
    close(dq, qr.grad, dt, 'attn_bwd', f32_mult=2.0)
    if dt == torch.bfloat16:
        # rows staged through LDS (whole-line requests) against the direct kernels: the same arithmetic on the same values
        with tuning.override(attn_staged=0):
            out0 = ops.attn_fwd(qkv, Fr, H, W, C, dh, ph, pw, window)
            dq0 = ops.attn_bwd(qkv, dout, Fr, H, W, C, dh, ph, pw, window)
        assert torch.equal(out.cpu(), out0.cpu()) and torch.equal(dq.cpu(), dq0.cpu())


AB_CASES = [  # F, H, W, ph, pw   (C = 64, dim_head 32: two heads)
    (2, 12, 20, 6, 10),            # L=60 (two 32-token blocks), 1 Mpx partition, 4 partitions per frame
    (1, 8, 20, 8, 10),             # L=80 (three blocks), Gen1 partition
    (3, 6, 10, 6, 10),             # one partition per frame; more partitions than one workgroup's waves
]


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('window,ln', [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize('case', AB_CASES)
def test_attn_block_fused(backend, dt, window, ln, case):
    """Fused attention half (csrc/attn_block.hpp) against an fp64 autograd restatement of maxvit.py:229,268,343-354."""
    Fr, H, W, ph, pw = case
    C, dh, eps = 64, 32, 1e-5
    assert ops.attn_block_supported(dt, C, dh, ph * pw)
    x = rnd((Fr, H, W, C), backend, dt, 1)
    dxm = rnd((Fr, H, W, C), backend, dt, 2)
    ln_w = (1.0 + 0.3 * rnd((C,), backend, torch.float32, 3)) if ln else None
    ln_b = 0.2 * rnd((C,), backend, torch.float32, 4) if ln else None
    wqkv32 = rnd((3 * C, C), backend, torch.float32, 5, C ** -0.5)
    bqkv = 0.3 * rnd((3 * C,), backend, torch.float32, 6)
    wp32 = rnd((C, C), backend, torch.float32, 7, C ** -0.5)
    bp = 0.3 * rnd((C,), backend, torch.float32, 8)
    gamma = 0.5 + rnd((C,), backend, torch.float32, 9).abs()
    wqkv, wp = wqkv32.to(dt), wp32.to(dt)
    wpg_t = (wp32 * gamma[:, None]).t().contiguous().to(dt)

    xmid, a = ops.attn_block_fwd(x, ln_w, ln_b, wqkv, bqkv, wp, bp, gamma, Fr, H, W, C, dh, ph, pw, window, eps, want_a=True)
    xmid2, a2 = ops.attn_block_fwd(x, ln_w, ln_b, wqkv, bqkv, wp, bp, gamma, Fr, H, W, C, dh, ph, pw, window, eps, want_a=False)
    assert a2 is None and torch.equal(xmid, xmid2)

    xr = f64(x).requires_grad_(True)
    lw = f64(ln_w).requires_grad_(True) if ln else None
    lb = f64(ln_b).requires_grad_(True) if ln else None
    u_r = F.layer_norm(xr, (C,), lw, lb, eps) if ln else xr
    u_r.retain_grad()
    qkv_r = u_r @ f64(wqkv).t() + f64(bqkv)
    qkv_r.retain_grad()
    a_r = ref_attention(qkv_r, Fr, H, W, C, dh, ph, pw, window)
    wpg = (f64(wpg_t).t() / f64(gamma)[:, None])          # = the (rounded) proj weight the backward kernel sees
    xmid_r = xr + f64(gamma) * (a_r @ f64(wp).t() + f64(bp))
    close(a, a_r, dt, 'attn_block a')
    close(xmid, xmid_r, dt, 'attn_block xmid')
    xmid_r.backward(f64(dxm))

    if ph * pw > 64:               # three 32-token blocks per partition: forward only (training takes the op-by-op chain)
        return
    dln_w = torch.zeros(C, dtype=torch.float32, device=backend) if ln else None
    dln_b = torch.zeros(C, dtype=torch.float32, device=backend) if ln else None
    dx, dqkv, u = ops.attn_block_bwd(x, dxm, ln_w, ln_b, wqkv, bqkv, wpg_t, dln_w, dln_b, Fr, H, W, C, dh, ph, pw, window, eps)
    close(dqkv, qkv_r.grad, dt, 'attn_block dqkv', f32_mult=2.0)
    close(dx, xr.grad, dt, 'attn_block dx', f32_mult=2.0)
    if ln:
        close(u, u_r, dt, 'attn_block u')
        close(dln_w, lw.grad, dt, 'attn_block dln_w', f32_mult=4.0)
        close(dln_b, lb.grad, dt, 'attn_block dln_b', f32_mult=4.0)
    else:
        assert u is None


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('window', [True, False])
@pytest.mark.parametrize('case', [c for c in AB_CASES if c[3] * c[4] <= 64])
def test_attn_block_bwd_preln(backend, dt, window, case):
    """rvt_attn_block_bwd_preln (a stage's first block: the backward carried through the down-sampling LayerNorm in front of it,
    maxvit.py:177 + :268) against fp64 autograd of LN -> attention half, and against the two launches it replaces
    (rvt_attn_block_bwd without norm1 + rvt_layernorm_bwd): same dqkv bits."""
    Fr, H, W, ph, pw = case
    C, dh, eps = 64, 32, 1e-5
    y0 = rnd((Fr, H, W, C), backend, dt, 1) * 1.5 + 0.25
    dxm = rnd((Fr, H, W, C), backend, dt, 2)
    ln_w = 1.0 + 0.3 * rnd((C,), backend, torch.float32, 3)
    ln_b = 0.2 * rnd((C,), backend, torch.float32, 4)
    wqkv32 = rnd((3 * C, C), backend, torch.float32, 5, C ** -0.5)
    bqkv = 0.3 * rnd((3 * C,), backend, torch.float32, 6)
    wp32 = rnd((C, C), backend, torch.float32, 7, C ** -0.5)
    bp = 0.3 * rnd((C,), backend, torch.float32, 8)
    gamma = 0.5 + rnd((C,), backend, torch.float32, 9).abs()
    wqkv, wp = wqkv32.to(dt), wp32.to(dt)
    wpg_t = (wp32 * gamma[:, None]).t().contiguous().to(dt)
    x = ops.layernorm_fwd(y0, ln_w, ln_b, eps)                        # the block input as the forward stored it

    yr = f64(y0).requires_grad_(True)
    lw, lb = f64(ln_w).requires_grad_(True), f64(ln_b).requires_grad_(True)
    # the kernels see the ROUNDED block input x; the norm's own backward sees y0: x_r = x + (LN(y0) - LN(y0).detach()) keeps both
    ln_r = F.layer_norm(yr, (C,), lw, lb, eps)
    x_r = f64(x) + (ln_r - ln_r.detach())
    qkv_r = x_r @ f64(wqkv).t() + f64(bqkv)
    qkv_r.retain_grad()
    a_r = ref_attention(qkv_r, Fr, H, W, C, dh, ph, pw, window)
    xmid_r = x_r + f64(gamma) * (a_r @ (f64(wpg_t).t() / f64(gamma)[:, None]).t() + f64(bp))
    xmid_r.backward(f64(dxm))

    z = lambda: torch.zeros(C, dtype=torch.float32, device=backend)
    dlw, dlb, dlw2, dlb2 = z(), z(), z(), z()
    dy0, dqkv = ops.attn_block_bwd_preln(x, y0, dxm, ln_w, wqkv, bqkv, wpg_t, dlw, dlb, Fr, H, W, C, dh, ph, pw, window, eps)
    close(dqkv, qkv_r.grad, dt, 'attn_block_preln dqkv', f32_mult=2.0)
    close(dy0, yr.grad, dt, 'attn_block_preln dy0', f32_mult=2.0)
    close(dlw, lw.grad, dt, 'attn_block_preln dln_w', f32_mult=4.0)
    close(dlb, lb.grad, dt, 'attn_block_preln dln_b', f32_mult=4.0)
    # the two launches it replaces (dx rounded to the storage type in between)
    dx, dqkv2, u = ops.attn_block_bwd(x, dxm, None, None, wqkv, bqkv, wpg_t, None, None, Fr, H, W, C, dh, ph, pw, window, eps)
    dy0_2 = ops.layernorm_bwd(y0, ln_w, dx, None, dlw2, dlb2, eps)
    assert u is None and torch.equal(dqkv, dqkv2)
    close(dy0, dy0_2.double(), dt, 'attn_block_preln dy0 vs two launches', f32_mult=2.0)
    close(dlw, dlw2.double(), dt, 'attn_block_preln dln_w vs two launches', f32_mult=4.0)
    close(dlb, dlb2.double(), dt, 'attn_block_preln dln_b vs two launches', f32_mult=4.0)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('M,C', [(150, 16), (70, 72)])
def test_lstm_cell(backend, dt, M, C):
    x, h = rnd((M, C), backend, dt, 1), rnd((M, C), backend, dt, 2)
    c = rnd((M, C), backend, torch.float32, 3)
    w = rnd((4 * C, 2 * C), backend, torch.float32, 4, 0.3)
    b = rnd((4 * C,), backend, torch.float32, 5, 0.3)
    perm = weights.lstm_gate_perm(C, backend)
    w_t = w.to(dt)
    h_out = torch.empty(M, C, dtype=dt, device=backend)
    c_out = torch.empty(M, C, device=backend)
    gates = torch.empty(M, 4 * C, dtype=dt, device=backend)
    ops.lstm_fwd(x, h, c, w_t[perm].contiguous(), b[perm].contiguous(), h_out, c_out, gates)
    xr, hr, cr = f64(x).requires_grad_(True), f64(h).requires_grad_(True), f64(c).requires_grad_(True)
    wr, br = f64(w_t).requires_grad_(True), f64(b)
    mix = torch.cat([xr, hr], 1) @ wr.t() + br
    fg, ig, og = torch.sigmoid(mix[:, :C]), torch.sigmoid(mix[:, C:2 * C]), torch.sigmoid(mix[:, 2 * C:3 * C])
    gg = torch.tanh(mix[:, 3 * C:])
    cn = fg * cr + ig * gg
    hn = og * torch.tanh(cn)
    close(c_out, cn, dt, 'lstm c')
    close(h_out, hn, dt, 'lstm h')
    close(gates, torch.cat([fg, ig, og, gg], 1), dt, 'lstm gates')
    # backward of one step with incoming dh (two parts) and dc
    dh_in, dh_rec = rnd((M, C), backend, dt, 6), rnd((M, C), backend, dt, 7)
    dc_rec = rnd((M, C), backend, torch.float32, 8)
    (hn * (f64(dh_in) + f64(dh_rec))).sum().backward(retain_graph=True)
    (cn * f64(dc_rec)).sum().backward()
    dz = torch.empty(M, 4 * C, dtype=dt, device=backend)
    dc_io = dc_rec.clone()
    # the kernel consumes the *saved* (rounded) gates and fp32 cell states
    ops.lstm_gates_bwd(dh_in, dh_rec, dc_io, gates, c_out, c, dz)
    close(dc_io, cr.grad, dt, 'lstm dc_prev', f32_mult=2.0)
    dx = torch.empty(M, C, dtype=dt, device=backend)
    dhp = torch.empty(M, C, dtype=dt, device=backend)
    ops.lstm_dgrad(dz, w_t.t().contiguous(), dx, dhp)
    close(dx, xr.grad, dt, 'lstm dx', f32_mult=2.0)
    close(dhp, hr.grad, dt, 'lstm dh_prev', f32_mult=2.0)
    ops.lstm_dgrad(dz, w_t.t().contiguous(), dx, dhp)
    dw = torch.zeros(4 * C, 2 * C, device=backend)
    dbias = torch.zeros(4 * C, device=backend)
    ops.lstm_wgrad(dz, x, h, dw, dbias)
    close(dw, wr.grad, dt, 'lstm dw', f32_mult=2.0)
    close(dbias, f64(dz).sum(0), dt, 'lstm dbias (fused column sum)', f32_mult=1.0)


CONV_CASES = [  # F, H, W, Cin, Cout, k, s, p
    (2, 8, 12, 16, 32, 3, 2, 1),
    (1, 16, 24, 20, 16, 7, 4, 3),     # stem: Cin=20 padded to 24
    (2, 8, 8, 16, 24, 2, 2, 0),       # non-overlapping patch (overlap=False)
    (1, 6, 10, 72, 136, 3, 2, 1),
    (2, 8, 32, 16, 32, 3, 2, 1),      # output width 16: the weight gradient's unit-linear im2col path, Cout <= 64 (transposed product)
    (1, 16, 64, 20, 72, 7, 4, 3),     # ... same with the stem geometry and Cout > 64 (direct product)
]


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv(backend, dt, case):
    Fr, H, W, Cin, Cout, k, s, p = case
    cp = weights.round8(Cin)
    x = rnd((Fr, H, W, cp), backend, dt, 1)
    x[..., Cin:] = 0
    w = rnd((Cout, Cin, k, k), backend, torch.float32, 2, 0.2).to(dt)
    y = ops.conv_fwd(x, weights.pack_conv_fwd(w.float(), cp, dt), k, s, p)
    xr = f64(x)[..., :Cin].permute(0, 3, 1, 2).requires_grad_(True)
    wr = f64(w).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    close(y, yr.permute(0, 2, 3, 1), dt, 'conv_fwd')
    dy = rnd(tuple(y.shape), backend, dt, 3)
    yr.backward(f64(dy).permute(0, 3, 1, 2))
    dw = torch.zeros(Cout, k * k * cp, device=backend)
    ops.conv_wgrad(x, dy, dw, k, s, p)
    close(weights.unpack_conv_wgrad(dw, Cin, k), wr.grad, dt, 'conv_wgrad')
    if Cin % 8 == 0:
        add = rnd((Fr, H, W, Cin), backend, dt, 4)
        din = ops.conv_dgrad(dy, weights.pack_conv_dgrad(w.float(), s, p, dt), add, H, W, Cin, k, s, p)
        close(din, xr.grad.permute(0, 2, 3, 1) + f64(add), dt, 'conv_dgrad')


STEM_CASES = [  # F, Cin, h, w, H, W  (uint8 planes h x w, zero padded to the model resolution H x W)
    (1, 20, 24, 40, 24, 40),          # one partial 32-pixel segment, row groups past Ho
    (2, 20, 30, 136, 32, 136),        # two segments (second partial), bottom padding rows (h < H)
    (1, 20, 16, 260, 16, 264),        # three segments, columns past the real width (w < W)
    (1, 3, 40, 132, 40, 132),         # few planes: odd number of contraction rows (21), 11 k-steps padded to 12
    (1, 20, 48, 136, 48, 136),        # has an interior item (second segment, rows 13..31): the unchecked load path
]


@pytest.mark.parametrize('case', STEM_CASES)
def test_stem(backend, case):
    """The stem kernels on the uint8 planes (csrc/stem.hpp) against fp64 conv2d + layer_norm of the cast + padded input
    (reference maxvit.py:160-177 on modules/detection.py:133-134, utils/padding.py:29-44) and against the prepack + GEMM route."""
    dt = torch.bfloat16
    Fr, Cin, h, w, H, W = case
    g = torch.Generator().manual_seed(5)
    src = torch.randint(0, 256, (Fr, Cin, h, w), generator=g, dtype=torch.uint8)
    src[:, :, ::3, ::5] = 0
    cp = weights.round8(Cin)
    wt = rnd((64, Cin, 7, 7), backend, torch.float32, 2, 0.02).to(dt)
    wp = weights.pack_conv_fwd(wt.float(), cp, dt)
    lw, lb = (1 + rnd((64,), backend, torch.float32, 3, 0.2)), rnd((64,), backend, torch.float32, 4, 0.2)
    srcd = src.to(backend)
    assert ops.stem_supported(srcd, dt, 64, 7, 4, 3)
    assert not ops.stem_supported(srcd.float(), dt, 64, 7, 4, 3) and not ops.stem_supported(srcd, torch.float32, 64, 7, 4, 3)
    y0, x = ops.stem_fwd(srcd, wp, lw, lb, H, W, 1e-5)
    xin = torch.zeros(Fr, Cin, H, W, dtype=torch.float64)
    xin[:, :, :h, :w] = src.double()
    wr = f64(wt).requires_grad_(True)
    yr = F.conv2d(xin, wr, None, 4, 3)
    close(y0, yr.permute(0, 2, 3, 1), dt, 'stem y0')
    # LayerNorm of the STORED (bf16) conv output: what the backward differentiates
    xr = F.layer_norm(f64(y0), (64,), f64(lw), f64(lb), 1e-5)
    close(x, xr, dt, 'stem LN(y0)')
    # same products as the GEMM route on the prepacked copy (uint8 is exact in bf16): only the summation order differs
    inp = ops.prepack_input(srcd, H, W, cp, dt)
    y_ref = ops.conv_fwd(inp, wp, 7, 4, 3)
    close(y0, f64(y_ref), dt, 'stem vs conv_fwd', f32_mult=0.5)
    dy = rnd(tuple(y0.shape), backend, dt, 6)
    yr.backward(f64(dy).permute(0, 3, 1, 2))
    dw = torch.full((64, 49 * cp), 0.5, device=backend)           # += semantics (gradient buckets accumulate)
    ops.stem_wgrad(srcd, dy, dw, H, W)
    got = weights.unpack_conv_wgrad(dw - 0.5, Cin, 7)
    close(got, wr.grad, dt, 'stem wgrad')
    pad_cols = (dw - 0.5).view(64, 49, cp)[:, :, Cin:]
    assert float(pad_cols.abs().max()) == 0.0 if cp > Cin else True


@pytest.mark.parametrize('case', [(3, 16, 24, 64, 128), (5, 24, 16, 128, 128), (9, 8, 16, 256, 128), (9, 8, 16, 384, 128),
                                  (9, 8, 16, 512, 128)])      # F, H, W, Cin, Cout
def test_conv_dgrad4(backend, case):
    """Input gradient of the 3x3 / 2 / 1 conv as one product over 2x2 pixel blocks (csrc/ppgemm.hpp GATHER) vs fp64 autograd and vs
    the four parity-class launches: image borders, ragged last row tile, with and without an added cotangent, the per-tile tap lists
    (Cin = 64: one N tile with all taps; 128: two tiles, 2 + 4 taps; 256: four tiles, 1 + 2 + 2 + 4 taps; 384 / 512: six / eight
    tiles - ADVICE r3: the tap table had four entries).  Cout < 128 is rejected (a single-tap tile needs two K tiles)."""
    assert not ops.conv_dgrad4_supported(torch.bfloat16, 16, 16, 128, 64, 3, 2, 1, 8)
    Fr, H, W, Cin, Cout = case
    dt = torch.bfloat16
    assert ops.conv_dgrad4_supported(dt, H, W, Cin, Cout, 3, 2, 1, Fr)
    w = rnd((Cout, Cin, 3, 3), backend, torch.float32, 2, 0.2).to(dt)
    dy = rnd((Fr, H // 2, W // 2, Cout), backend, dt, 3)
    add = rnd((Fr, H, W, Cin), backend, dt, 4)
    xr = torch.zeros(Fr, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, f64(w), None, 2, 1).backward(f64(dy).permute(0, 3, 1, 2))
    want = xr.grad.permute(0, 2, 3, 1)
    wd4 = weights.pack_conv_dgrad4(w.float(), dt)
    close(ops.conv_dgrad4(dy, wd4, None, H, W, Cin), want, dt, 'conv_dgrad4')
    close(ops.conv_dgrad4(dy, wd4, add, H, W, Cin), want + f64(add), dt, 'conv_dgrad4 + add')
    old = ops.conv_dgrad(dy, weights.pack_conv_dgrad(w.float(), 2, 1, dt), add, H, W, Cin, 3, 2, 1)
    close(ops.conv_dgrad4(dy, wd4, add, H, W, Cin), f64(old), dt, 'conv_dgrad4 vs parity-class route', f32_mult=0.5)


@pytest.mark.parametrize('case', [(3, 16, 24, 64, 256), (2, 20, 36, 128, 512), (5, 12, 20, 256, 512), (9, 8, 16, 64, 256)])   # F, H, W, Cin, Cout
def test_conv_fwd_pp(backend, case):
    """The 3x3 / 2 / 1 down-sampling conv on the 256-wide kernel with the im2col gather in its load stream (csrc/ppgemm.hpp GATHER = 2)
    vs fp64 conv2d and vs the 128-row engine: image borders (top / left taps outside), ragged last row tile, several tiles per
    workgroup, one and two N tiles, 9 / 18 / 36 K tiles."""
    Fr, H, W, Cin, Cout = case
    dt = torch.bfloat16
    x = rnd((Fr, H, W, Cin), backend, dt, 1)
    w = rnd((Cout, Cin, 3, 3), backend, torch.float32, 2, 0.2).to(dt)
    wp = weights.pack_conv_fwd(w.float(), Cin, dt)
    want = F.conv2d(f64(x).permute(0, 3, 1, 2), f64(w), None, 2, 1).permute(0, 2, 3, 1)
    with tuning.override(conv_fwd_pp=1):
        y = ops.conv_fwd(x, wp, 3, 2, 1)
    with tuning.override(conv_fwd_pp=0):
        y0 = ops.conv_fwd(x, wp, 3, 2, 1)
    close(y, want, dt, 'conv_fwd_pp')
    close(y, f64(y0), dt, 'conv_fwd_pp vs the 128-row engine', f32_mult=0.5)


CONV_WGRAD_TN_CASES = [  # F, H, W, Cin, Cout, k, stride, pad
    (3, 24, 40, 128, 256, 3, 2, 1),      # stage-3 down-sampling conv: K = 1152 = 4.5 k tiles (the last one half beyond K), two taps per tile
    (5, 12, 20, 256, 512, 3, 2, 1),      # stage 4: one tap per k tile, two n tiles, Wo = 10 (a 64-token step spans 6 image rows and frames)
    (2, 16, 24, 64, 256, 3, 1, 1),       # PAFPN 3x3 / 1: four taps per tile, K = 576 (2.25 tiles)
    (2, 18, 22, 256, 256, 1, 1, 0),      # 1x1: a plain token contraction through the gather path
    (3, 26, 38, 64, 256, 2, 2, 0),       # non-overlapping 2x2 / 2 (overlap = False configs), odd-sized remainder rows
]


@pytest.mark.parametrize('case', CONV_WGRAD_TN_CASES)
def test_conv_wgrad_tn(backend, case):
    """Conv weight gradient on the 256-wide token-contraction kernel with im2col as the LDS-DMA source address (csrc/ppgemm_tn.hpp,
    CONV) vs fp64 autograd and vs the split-K im2col engine: image borders, token slices that end inside an image row, a last k
    tile partly beyond K, accumulation into a non-zero dw."""
    Fr, H, W, Cin, Cout, k, s, p = case
    dt = torch.bfloat16
    x = rnd((Fr, H, W, Cin), backend, dt, 1)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = rnd((Fr, Ho, Wo, Cout), backend, dt, 3)
    wr = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(f64(x).permute(0, 3, 1, 2), wr, None, s, p).backward(f64(dy).permute(0, 3, 1, 2))
    dw = torch.zeros(Cout, k * k * Cin, device=backend)
    ops.conv_wgrad(x, dy, dw, k, s, p)
    close(weights.unpack_conv_wgrad(dw, Cin, k), wr.grad, dt, 'conv_wgrad (ppgemm_tn)')
    with tuning.override(conv_wgrad_tn=0):
        dw0 = torch.zeros(Cout, k * k * Cin, device=backend)
        ops.conv_wgrad(x, dy, dw0, k, s, p)
    close(dw, dw0.double(), dt, 'conv_wgrad ppgemm_tn vs split-K engine', f32_mult=0.5)
    ops.conv_wgrad(x, dy, dw, k, s, p)                    # accumulates
    close(weights.unpack_conv_wgrad(dw, Cin, k), 2 * wr.grad, dt, 'conv_wgrad (ppgemm_tn) accumulate', f32_mult=2.0)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('N,H,W,C', [(2, 5, 7, 16), (1, 6, 4, 48), (3, 3, 3, 8)])
def test_dwconv(backend, dt, N, H, W, C):
    x, dy = rnd((N, H, W, C), backend, dt, 1), rnd((N, H, W, C), backend, dt, 2)
    w, b = rnd((C, 9), backend, torch.float32, 3, 0.3), rnd((C,), backend, torch.float32, 4)
    y = ops.dwconv(x, w, b, 3)
    xr = f64(x).permute(0, 3, 1, 2).requires_grad_(True)
    wr, br = f64(w).reshape(C, 1, 3, 3).requires_grad_(True), f64(b).requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1, groups=C)
    close(y, yr.permute(0, 2, 3, 1), dt, 'dwconv fwd')
    yr.backward(f64(dy).permute(0, 3, 1, 2))
    dx = ops.dwconv(dy, w, None, 3, transpose=True)
    close(dx, xr.grad.permute(0, 2, 3, 1), dt, 'dwconv bwd input')
    dw, db = torch.zeros(C, 9, device=backend), torch.zeros(C, device=backend)
    ops.dwconv_wgrad(x, dy, dw, db, 3)
    close(dw, wr.grad.reshape(C, 9), dt, 'dwconv bwd weight')
    close(db, br.grad, dt, 'dwconv bwd bias')


@pytest.mark.parametrize('dt', DTYPES)
def test_state_reset(backend, dt):
    st = rnd((3, 4, 5, 8), backend, dt, 1)
    ref = st.clone()
    ops.state_reset_masked(st, torch.tensor([True, False, True]))
    ref[0] = 0
    ref[2] = 0
    assert torch.equal(st.cpu(), ref.cpu())


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M', [130, 1000, 31, 33, 63, 65, 129, 257])
def test_mlp_bwd_recompute(backend, dt, M):
    """Recompute backward of the MLP half (C = 64) vs fp64 autograd: rvt_mlp_bwd_recompute_dgrad + rvt_mlp_bwd_recompute_wgrad
    (the register-chained kernels of csrc/mlp_chain.hpp; what the stage driver calls)."""
    C = 64
    assert ops.mlp_bwd_fused_supported(dt, C)
    x = rnd((M, C), backend, dt, 1, 1.5)
    lw, lb = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0, rnd((C,), backend, torch.float32, 3, 0.2)
    w1, b1 = rnd((4 * C, C), backend, dt, 4, 0.2), rnd((4 * C,), backend, torch.float32, 5, 0.2)
    w2, b2 = rnd((C, 4 * C), backend, dt, 6, 0.1), rnd((C,), backend, torch.float32, 7, 0.2)
    gam = rnd((C,), backend, torch.float32, 8)
    dy = rnd((M, C), backend, dt, 9)

    xr = f64(x).requires_grad_(True)
    lwr, lbr = f64(lw).requires_grad_(True), f64(lb).requires_grad_(True)
    w1r, b1r = f64(w1).requires_grad_(True), f64(b1).requires_grad_(True)
    v2 = F.layer_norm(xr, (C,), lwr, lbr, 1e-5)
    g = F.gelu(v2 @ w1r.t() + b1r)
    g.retain_grad()
    y2 = g @ f64(w2).t() + f64(b2)                       # the LayerScale branch: xout = x + gamma * y2
    want = xr + f64(gam) * y2
    want.backward(f64(dy))

    w2g_t = (f64(w2) * f64(gam)[:, None]).t().to(dt).contiguous().to(backend)
    w1_t = f64(w1).t().to(dt).contiguous().to(backend)
    z = lambda *s: torch.zeros(*s, device=backend)
    dlw, dlb, dw1, db1, s2, cs2 = z(C), z(C), z(4 * C, C), z(4 * C), z(C, 4 * C), z(C)
    def run():
        d = ops.mlp_bwd_recompute_dgrad(dy, x, lw, lb, w1, b1, w2g_t, w1_t, dlw, dlb, 1e-5)
        ops.mlp_bwd_recompute_wgrad(dy, x, lw, lb, w1, b1, w2g_t, dw1, db1, s2, cs2, 1e-5)
        return d
    dxm = run()
    close(dxm, xr.grad, dt, 'mlp_bwd_fused dxmid')
    close(dlw, lwr.grad, dt, 'mlp_bwd_fused dln_w', f32_mult=2.0)
    close(dlb, lbr.grad, dt, 'mlp_bwd_fused dln_b', f32_mult=2.0)
    close(dw1, w1r.grad, dt, 'mlp_bwd_fused dW1', f32_mult=2.0)
    close(db1, b1r.grad, dt, 'mlp_bwd_fused db1', f32_mult=2.0)
    # raw fc2 products: S2 = dy^T g, cs2 = colsum(dy)  (gamma is applied by the LayerScale fold)
    close(s2, f64(dy).t() @ g.detach(), dt, 'mlp_bwd_fused S2', f32_mult=2.0)
    close(cs2, f64(dy).sum(0), dt, 'mlp_bwd_fused cs2')
    # accumulation semantics (+=) of every parameter-gradient output
    dxm2 = run()
    assert torch.equal(dxm2.cpu(), dxm.cpu())
    close(dw1, 2 * w1r.grad, dt, 'mlp_bwd_fused dW1 accumulate', f32_mult=2.0)
    close(cs2, 2 * f64(dy).sum(0), dt, 'mlp_bwd_fused cs2 accumulate')


@pytest.mark.parametrize('M', [130, 1000, 5000, 31, 33, 63, 65, 129, 257])
def test_mlp_bwd_recompute_both(backend, M):
    """rvt_mlp_bwd_recompute_both (bf16, C = 64: weight gradients AND input gradient from one recompute, one launch) vs fp64 autograd
    and vs the two-launch route (the input gradient passes dh W1 through a bf16 tile: close, not bit-equal)."""
    C, dt = 64, torch.bfloat16
    with tuning.override(route_mlp_bwd_both=1, mlp_chain_wgrad=1, mlp_chain=1):
        if not ops.mlp_bwd_both_supported(dt, C):
            pytest.skip('chain MLP kernels routed off on this backend')
        x = rnd((M, C), backend, dt, 1, 1.5)
        lw, lb = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0, rnd((C,), backend, torch.float32, 3, 0.2)
        w1, b1 = rnd((4 * C, C), backend, dt, 4, 0.2), rnd((4 * C,), backend, torch.float32, 5, 0.2)
        w2 = rnd((C, 4 * C), backend, dt, 6, 0.1)
        gam = rnd((C,), backend, torch.float32, 8)
        dy = rnd((M, C), backend, dt, 9)
        xr = f64(x).requires_grad_(True)
        lwr, lbr = f64(lw).requires_grad_(True), f64(lb).requires_grad_(True)
        w1r, b1r = f64(w1).requires_grad_(True), f64(b1).requires_grad_(True)
        g = F.gelu(F.layer_norm(xr, (C,), lwr, lbr, 1e-5) @ w1r.t() + b1r)
        (xr + f64(gam) * (g @ f64(w2).t())).backward(f64(dy))
        w2g_t = (f64(w2) * f64(gam)[:, None]).t().to(dt).contiguous().to(backend)
        w1_t = f64(w1).t().to(dt).contiguous().to(backend)
        z = lambda *s: torch.zeros(*s, device=backend)
        dlw, dlb, dw1, db1, s2, cs2 = z(C), z(C), z(4 * C, C), z(4 * C), z(C, 4 * C), z(C)
        dxm = ops.mlp_bwd_recompute_both(dy, x, lw, lb, w1, b1, w2g_t, w1_t, dlw, dlb, dw1, db1, s2, cs2, 1e-5)
        close(dxm, xr.grad, dt, 'mlp_bwd_both dxmid', f32_mult=2.0)
        close(dlw, lwr.grad, dt, 'mlp_bwd_both dln_w', f32_mult=4.0)
        close(dlb, lbr.grad, dt, 'mlp_bwd_both dln_b', f32_mult=4.0)
        close(dw1, w1r.grad, dt, 'mlp_bwd_both dW1', f32_mult=4.0)
        close(db1, b1r.grad, dt, 'mlp_bwd_both db1', f32_mult=4.0)
        close(s2, f64(dy).t() @ g.detach(), dt, 'mlp_bwd_both S2', f32_mult=4.0)
        close(cs2, f64(dy).sum(0), dt, 'mlp_bwd_both cs2', f32_mult=2.0)
        # against the two-launch route: weight-gradient side identical (same code), input gradient within bf16 rounding
        dlw2, dlb2, dw12, db12, s22, cs22 = z(C), z(C), z(4 * C, C), z(4 * C), z(C, 4 * C), z(C)
        d2 = ops.mlp_bwd_recompute_dgrad(dy, x, lw, lb, w1, b1, w2g_t, w1_t, dlw2, dlb2, 1e-5)
        ops.mlp_bwd_recompute_wgrad(dy, x, lw, lb, w1, b1, w2g_t, dw12, db12, s22, cs22, 1e-5)
        assert torch.equal(dw1.cpu(), dw12.cpu()) and torch.equal(s2.cpu(), s22.cpu()) and torch.equal(cs2.cpu(), cs22.cpu())
        close(dxm, d2.double(), dt, 'mlp_bwd_both dxmid vs the two-launch route')
        # accumulation semantics and run-to-run reproducibility of the input gradient
        dxm3 = ops.mlp_bwd_recompute_both(dy, x, lw, lb, w1, b1, w2g_t, w1_t, dlw, dlb, dw1, db1, s2, cs2, 1e-5)
        assert torch.equal(dxm3.cpu(), dxm.cpu())
        close(dw1, 2 * w1r.grad, dt, 'mlp_bwd_both dW1 accumulate', f32_mult=4.0)
        close(dlw, 2 * lwr.grad, dt, 'mlp_bwd_both dln_w accumulate', f32_mult=4.0)


@pytest.mark.parametrize('dt', DTYPES)
def test_gather_frames(backend, dt):
    """rvt_gather_frames (labelled-frame gather of the training step, modules/utils/detection.py:32-46) and its backward."""
    frames = rnd((7, 3, 5, 16), backend, dt, 1).requires_grad_(True)
    idx = torch.tensor([5, 0, 6, 2], dtype=torch.int32, device=backend)
    out = ops.gather_frames(frames, idx)
    assert torch.equal(out.detach().cpu(), frames.detach().cpu()[idx.cpu().long()])
    cot = rnd(tuple(out.shape), backend, dt, 2)
    out.backward(cot)
    want = torch.zeros_like(frames.detach()).cpu()
    want[idx.cpu().long()] = cot.cpu()
    assert torch.equal(frames.grad.cpu(), want)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('M,resident', [(1000, 2), (300, 0), (2049, 3), (31, 0), (33, 0), (63, 0), (65, 2), (129, 0), (257, 2)])
def test_mlp_stream_fwd(backend, dt, M, resident):
    """Streamed-weight chain forward of the MLP half at C = 128 (csrc/mlp_stream.hpp: weights by LDS-DMA in hidden chunks, nothing
    saved) vs fp64 autograd and vs the LDS-staged kernel it replaces; `resident` workgroups so that a workgroup walks several
    tiles (deferred row stores, weight stream across tile boundaries) and the last tile is ragged."""
    C = 128
    x = rnd((M, C), backend, dt, 1, 1.5)
    lw, lb = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0, rnd((C,), backend, torch.float32, 3, 0.2)
    w1, b1 = rnd((4 * C, C), backend, dt, 4, 0.2), rnd((4 * C,), backend, torch.float32, 5, 0.2)
    w2, b2 = rnd((C, 4 * C), backend, dt, 6, 0.1), rnd((C,), backend, torch.float32, 7, 0.2)
    gam = rnd((C,), backend, torch.float32, 8)
    with tuning.override(mlp_stream=1, chain_resident=resident):
        y = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=False)[0]
    v2 = F.layer_norm(f64(x), (C,), f64(lw), f64(lb), 1e-5)
    want = f64(x) + f64(gam) * (F.gelu(v2 @ f64(w1).t() + f64(b1)) @ f64(w2).t() + f64(b2))
    close(y, want, dt, 'mlp_fwd streamed')
    if dt == torch.bfloat16:
        with tuning.override(mlp_stream=0):
            y0 = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=False)[0]
        close(y, y0.double(), dt, 'mlp_fwd streamed vs LDS-staged')


@pytest.mark.parametrize('M,resident,want_u', [(1000, 2, True), (300, 0, True), (2049, 3, False), (31, 0, True), (33, 0, True), (63, 0, False),
                                               (65, 2, True), (129, 0, True), (257, 2, True)])
def test_ln_linear_fwd(backend, M, resident, want_u):
    """norm1 + qkv projection of a C = 128 block in one launch (csrc/ln_linear.hpp, bf16) vs fp64 and vs the two launches it
    replaces; `resident` workgroups so that a wave walks several tiles (row prefetch across tiles) and the last tile is ragged;
    want_u = False is the no-grad forward (u not kept) and must give the same y bit for bit."""
    C, N, dt = 128, 384, torch.bfloat16
    assert ops.ln_linear_supported(dt, C, N) and not ops.ln_linear_supported(torch.float32, C, N)
    x = rnd((M, C), backend, dt, 1, 1.5)
    lw, lb = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0, rnd((C,), backend, torch.float32, 3, 0.2)
    w, b = rnd((N, C), backend, dt, 4, 0.2), rnd((N,), backend, torch.float32, 5, 0.2)
    with tuning.override(chain_resident=resident):
        u, y = ops.ln_linear_fwd(x, lw, lb, w, b, 1e-5, want_u=want_u)
        y_again = ops.ln_linear_fwd(x, lw, lb, w, b, 1e-5, want_u=not want_u)[1]
    assert torch.equal(y.cpu(), y_again.cpu())
    u0 = ops.layernorm_fwd(x, lw, lb, 1e-5)
    y0 = ops.linear_fwd(u0, w, b)
    if want_u:
        close(u, F.layer_norm(f64(x), (C,), f64(lw), f64(lb), 1e-5), dt, 'ln_linear u')
        assert (u.float() - u0.float()).abs().max().item() <= 2 ** -6 * u0.float().abs().max().item()      # one bf16 ulp
    else:
        assert u is None
    # y against the fp64 product of the ROUNDED u (what both routes multiply), then against the op-by-op route
    ur = f64(u if want_u else u0)
    close(y, ur @ f64(w).t() + f64(b), dt, 'ln_linear y')
    close(y, y0.double(), dt, 'ln_linear y vs layernorm_fwd + linear_fwd', f32_mult=2.0)
    # no LayerNorm (first block of a stage): the plain product through the same kernel
    un, yn = ops.ln_linear_fwd(x, None, None, w, b, 1e-5, want_u=want_u)
    assert un is None
    close(yn, f64(x) @ f64(w).t() + f64(b), dt, 'ln_linear y (no LayerNorm)')
    close(yn, ops.linear_fwd(x, w, b).double(), dt, 'ln_linear y (no LayerNorm) vs linear_fwd', f32_mult=2.0)
    with tuning.override(ln_linear=0):
        assert not ops.ln_linear_supported(dt, C, N)


@pytest.mark.parametrize('M,resident,save', [(700, 16, True), (300, 0, True), (1031, 24, False), (31, 0, True), (2500, 64, True), (33, 0, True),
                                             (63, 0, True), (65, 16, False), (129, 0, True), (257, 16, True)])
def test_linear_gelu_weight_stationary(backend, M, resident, save):
    """fc1 + GELU (+ GELU') of a C = 256 block on the weight-stationary kernel (csrc/ln_linear.hpp lin_gelu_ws_kernel, bf16, behind
    rvt_linear_gelu_fwd at K = 256, N = 1024) vs fp64 and vs the GEMM engine it replaces; `resident` workgroups = 8 column groups x
    token streams (2, 3 and 8 streams: both workgroup -> (group, stream) mappings), several tiles per wave, ragged last tile;
    save = False is the no-grad forward (gp not kept, same g bit for bit)."""
    K, N, dt = 256, 1024, torch.bfloat16
    x = rnd((M, K), backend, dt, 1, 1.0)
    w, b = rnd((N, K), backend, dt, 4, 0.15), rnd((N,), backend, torch.float32, 5, 0.2)
    with tuning.override(chain_resident=resident, ln_linear=1):
        g, gp = ops.linear_gelu_fwd(x, w, b, want_grad=save)
        g_again = ops.linear_gelu_fwd(x, w, b, want_grad=not save)[0]
    assert torch.equal(g.cpu(), g_again.cpu())
    with tuning.override(ln_linear=0):
        g0, gp0 = ops.linear_gelu_fwd(x, w, b, want_grad=True)
    h = f64(x) @ f64(w).t() + f64(b)
    close(g, F.gelu(h), dt, 'linear_gelu (weight-stationary) g')
    close(g, g0.double(), dt, 'linear_gelu (weight-stationary) g vs GEMM engine', f32_mult=2.0)
    if save:
        hr = h.clone().requires_grad_(True)
        F.gelu(hr).sum().backward()
        close(gp, hr.grad, dt, "linear_gelu (weight-stationary) gp")
        close(gp, gp0.double(), dt, "linear_gelu (weight-stationary) gp vs GEMM engine", f32_mult=2.0)
    else:
        assert gp is None


def _mlp_case(backend, dt, M, C=128):
    x = rnd((M, C), backend, dt, 1, 1.5)
    lw, lb = rnd((C,), backend, torch.float32, 2) * 0.3 + 1.0, rnd((C,), backend, torch.float32, 3, 0.2)
    w1, b1 = rnd((4 * C, C), backend, dt, 4, 0.2), rnd((4 * C,), backend, torch.float32, 5, 0.2)
    w2, b2 = rnd((C, 4 * C), backend, dt, 6, 0.1), rnd((C,), backend, torch.float32, 7, 0.2)
    gam = rnd((C,), backend, torch.float32, 8) * 0.5 + 1.0
    dy = rnd((M, C), backend, dt, 9)
    xr = f64(x).requires_grad_(True)
    lwr, lbr = f64(lw).requires_grad_(True), f64(lb).requires_grad_(True)
    w1r, b1r = f64(w1).requires_grad_(True), f64(b1).requires_grad_(True)
    v2 = F.layer_norm(xr, (C,), lwr, lbr, 1e-5)
    g = F.gelu(v2 @ w1r.t() + b1r)
    want = xr + f64(gam) * (g @ f64(w2).t() + f64(b2))
    want.backward(f64(dy))
    w2g_t = (f64(w2) * f64(gam)[:, None]).t().to(dt).contiguous().to(backend)
    w1_t = f64(w1).t().to(dt).contiguous().to(backend)
    return dict(x=x, lw=lw, lb=lb, w1=w1, b1=b1, w2=w2, b2=b2, gam=gam, dy=dy, w2g_t=w2g_t, w1_t=w1_t, xr=xr, lwr=lwr, lbr=lbr,
                w1r=w1r, b1r=b1r, g=g.detach(), want=want.detach())


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('M,resident', [(1000, 2), (300, 0), (2049, 3), (31, 0), (33, 0), (63, 0), (65, 2), (129, 0), (257, 2)])
def test_mlp_stream_bwd_dgrad(backend, dt, M, resident):
    """Streamed-weight recompute backward of the MLP half at C = 128, input-gradient kernel (csrc/mlp_stream.hpp) vs fp64 autograd."""
    c = _mlp_case(backend, dt, M)
    C = 128
    dlw, dlb = torch.zeros(C, device=backend), torch.zeros(C, device=backend)
    with tuning.override(mlp_stream=1, chain_resident=resident):
        dxm = ops.mlp_bwd_recompute_dgrad(c['dy'], c['x'], c['lw'], c['lb'], c['w1'], c['b1'], c['w2g_t'], c['w1_t'], dlw, dlb, 1e-5)
    close(dxm, c['xr'].grad, dt, 'mlp_stream dxmid')
    close(dlw, c['lwr'].grad, dt, 'mlp_stream dln_w', f32_mult=2.0)
    close(dlb, c['lbr'].grad, dt, 'mlp_stream dln_b', f32_mult=2.0)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('M,grid', [(1000, 6), (300, 0), (2049, 4), (33, 0), (31, 0), (63, 0), (65, 2), (129, 0), (257, 2)])
def test_mlp_stream_bwd_wgrad(backend, dt, M, grid):
    """Streamed recompute backward of the MLP half at C = 128, weight-gradient kernel (csrc/mlp_stream.hpp: weight-stationary, two
    workgroups = hidden halves per tile stream, tiles by LDS-DMA, LayerNorm in place) vs fp64 autograd; `grid` = one_per_cu_grid
    (2 workgroups per stream) so that a stream walks several tiles; accumulation into existing buffers.  fp32 (parity twin of the
    route): the same entry point recomputes LN2 / fc1 / GELU / GELU' / dh with the op-by-op kernels."""
    C = 128
    c = _mlp_case(backend, dt, M)
    with tuning.override(mlp_stream=1, one_per_cu_grid=grid):
        assert ops.mlp_bwd_fused_supported(dt, C)
        dw1, db1 = torch.zeros(4 * C, C, device=backend), torch.zeros(4 * C, device=backend)
        s2, cs2 = torch.zeros(C, 4 * C, device=backend), torch.zeros(C, device=backend)
        for _ in range(2):
            ops.mlp_bwd_recompute_wgrad(c['dy'], c['x'], c['lw'], c['lb'], c['w1'], c['b1'], c['w2g_t'], dw1, db1, s2, cs2, 1e-5)
    close(dw1, 2 * c['w1r'].grad, dt, 'mlp_stream dW1', f32_mult=2.0)
    close(db1, 2 * c['b1r'].grad, dt, 'mlp_stream db1', f32_mult=2.0)
    close(s2, 2 * f64(c['dy']).t() @ c['g'], dt, 'mlp_stream S2', f32_mult=2.0)
    close(cs2, 2 * f64(c['dy']).sum(0), dt, 'mlp_stream cs2')
