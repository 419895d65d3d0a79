"""GPU micro-benchmark of rvt_linear_dgrad_ln (csrc/dgrad_ln.hpp) against the two launches it replaces, at the RVT-Base stage-2
shapes (C = 128, 1.9 M tokens; K = 512: fc1 / norm2, K = 384: qkv / norm1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops
dev, dt = torch.device('cuda', 0), torch.bfloat16


def timeit(fn, n=7):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, C, K in ((1935360, 128, 512), (1935360, 128, 384), (215040, 64, 192)):
    x, dy, dres = torch.randn(M, C, device=dev).to(dt), torch.randn(M, K, device=dev).to(dt), torch.randn(M, C, device=dev).to(dt)
    w, lw = (torch.randn(K, C, device=dev) * 0.1).to(dt), torch.rand(C, device=dev) + 0.5
    wt = w.t().contiguous()
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    t_f = timeit(lambda: ops.linear_dgrad_ln(dy, w, x, dres, lw, dw, db, 1e-5, out=out))
    du = torch.empty_like(x)
    t_g = timeit(lambda: ops.linear_dgrad(dy, wt, out=du))
    t_l = timeit(lambda: ops.layernorm_bwd(x, lw, du, dres, dw, db, 1e-5, out=out))
    dw.zero_(); db.zero_()
    a = ops.linear_dgrad_ln(dy, w, x, dres, lw, dw, db, 1e-5)
    dw2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    b = ops.layernorm_bwd(x, lw, ops.linear_dgrad(dy, wt), dres, dw2, db2, 1e-5)
    err = (a.float() - b.float()).abs().max().item() / b.float().abs().max().item()
    errw = (dw - dw2).abs().max().item() / dw2.abs().max().item()
    gb = (M * (K + 3 * C) * 2) / 1e9
    print(f'M={M} C={C} K={K}: fused {t_f:.3f} ms ({gb / t_f:.2f} TB/s) | GEMM {t_g:.3f} + LayerNorm bwd {t_l:.3f} = {t_g + t_l:.3f} ms   '
          f'dx diff {err:.1e}  dln_w diff {errw:.1e}', flush=True)
