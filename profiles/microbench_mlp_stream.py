"""GPU micro-benchmark of the C = 128 MLP half (stage 2 of RVT-Base: 1.94 M tokens; stage 3 of RVT-Tiny): the streamed-weight chain
kernels of csrc/mlp_stream.hpp (nothing saved, recompute backward) against the route they replace (LDS-staged forward saving
GELU / GELU' / LN2 + op-by-op backward)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops, tuning

dev, dt = torch.device('cuda', 0), torch.bfloat16
C = 128
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1935360


def timeit(fn, n=11):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[n // 2], ts[0]


g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
x, dy = rn(M, C).to(dt), rn(M, C).to(dt)
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
w1, w2 = rn(4 * C, C, sc=0.1).to(dt), rn(C, 4 * C, sc=0.1).to(dt)
b1, b2, gam = rn(4 * C, sc=0.1), rn(C, sc=0.1), torch.ones(C, device=dev)
w2gt = (w2.float() * gam[:, None]).t().contiguous().to(dt)
w1t = w1.t().contiguous()
gf = 16.0 * M * C * C * 1e-9
fwd = lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5)
with tuning.override(mlp_stream=1):
    t_new = timeit(fwd)
with tuning.override(mlp_stream=0):
    t_old = timeit(fwd)
    t_train = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=True, want_v2=True))
print(f'C={C} M={M} forward ({gf:.0f} GFLOP): streamed {t_new[0]:.3f} (min {t_new[1]:.3f}) ms = {gf / t_new[0]:.0f} TFLOP/s | '
      f'LDS-staged nothing saved {t_old[0]:.3f} | LDS-staged saving g, gp, v2 {t_train[0]:.3f} ms', flush=True)
if True:
    dlw, dlb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dw1, db1 = torch.zeros(4 * C, C, device=dev), torch.zeros(4 * C, device=dev)
    s2, cs2 = torch.zeros(C, 4 * C, device=dev), torch.zeros(C, device=dev)
    t_d = timeit(lambda: ops.mlp_bwd_recompute_dgrad(dy, x, lw, lb, w1, b1, w2gt, w1t, dlw, dlb, 1e-5))
    print(f'   recompute backward: dgrad {t_d[0]:.3f} (min {t_d[1]:.3f}) ms = {1.5 * gf / t_d[0]:.0f} TFLOP/s executed', flush=True)
    if ops.mlp_bwd_fused_supported(dt, C):
        t_w = timeit(lambda: ops.mlp_bwd_recompute_wgrad(dy, x, lw, lb, w1, b1, w2gt, dw1, db1, s2, cs2, 1e-5))
        print(f'   recompute backward: wgrad {t_w[0]:.3f} (min {t_w[1]:.3f}) ms = {2.0 * gf / t_w[0]:.0f} TFLOP/s executed', flush=True)
# the op-by-op backward it replaces (saved g, gp, v2)
with tuning.override(mlp_stream=0):
    _, hg, hgp, v2 = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=True, want_v2=True)
    s2, cs2 = torch.zeros(C, 4 * C, device=dev), torch.zeros(C, device=dev)
    dw1, db1 = torch.zeros(4 * C, C, device=dev), torch.zeros(4 * C, device=dev)
    dlw, dlb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t1 = timeit(lambda: ops.linear_wgrad(dy, hg, s2, colsum_out=cs2))
    t2 = timeit(lambda: ops.linear_dgrad(dy, w2gt.t().contiguous().t() if False else w2gt, mul=hgp))
    dhd = ops.linear_dgrad(dy, w2gt, mul=hgp)
    t3 = timeit(lambda: ops.linear_wgrad(dhd, v2, dw1, colsum_out=db1))
    t4 = timeit(lambda: ops.linear_dgrad_ln(dhd, w1, x, dy, lw, dlw, dlb, 1e-5))
    print(f'   op-by-op backward: fc2 wgrad {t1[0]:.3f} + fc2 dgrad*gp {t2[0]:.3f} + fc1 wgrad {t3[0]:.3f} + fc1 dgrad+LN {t4[0]:.3f} = '
          f'{t1[0] + t2[0] + t3[0] + t4[0]:.3f} ms', flush=True)
