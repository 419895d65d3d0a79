"""GPU micro-benchmark of the recompute MLP route at the RVT-Base 1Mpx stage-1 shape (C = 64, 7.74 M tokens):
forward with nothing saved, input-gradient half, weight-gradient half.  RVT_MLP_CHAIN=0 times the LDS-staged kernels of
csrc/mlp.hpp, default the register-chained kernels of csrc/mlp_chain.hpp."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

dev = torch.device('cuda', 0)
dt = torch.bfloat16
C, M = 64, 7741440


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, C, device=dev, generator=g).to(dt)
dy = torch.randn(M, C, device=dev, generator=g).to(dt)
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
w1 = (torch.randn(4 * C, C, device=dev, generator=g) * 0.1).to(dt)
w2 = (torch.randn(C, 4 * C, device=dev, generator=g) * 0.1).to(dt)
w2gt, w1t = w2.t().contiguous(), w1.t().contiguous()
b1, b2, gam = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
z = lambda *s: torch.zeros(*s, device=dev)
dlw, dlb, dw1, db1, s2, cs2 = z(C), z(C), z(4 * C, C), z(4 * C), z(C, 4 * C), z(C)
t_f = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=False))
t_d = timeit(lambda: ops.mlp_bwd_recompute_dgrad(dy, x, lw, lb, w1, b1, w2gt, w1t, dlw, dlb, 1e-5))
t_w = timeit(lambda: ops.mlp_bwd_recompute_wgrad(dy, x, lw, lb, w1, b1, w2gt, dw1, db1, s2, cs2, 1e-5))
t_b = timeit(lambda: ops.mlp_bwd_recompute_both(dy, x, lw, lb, w1, b1, w2gt, w1t, dlw, dlb, dw1, db1, s2, cs2, 1e-5)) if ops.mlp_bwd_both_supported(x.dtype, C) else float('nan')
print(f'   one-launch backward (rvt_mlp_bwd_recompute_both): {t_b:.3f} ms against {t_d + t_w:.3f} ms for dgrad + wgrad', flush=True)
print(f'RVT_MLP_CHAIN={os.environ.get("RVT_MLP_CHAIN", "1")} C={C} M={M}: fwd (nothing saved) {t_f:.3f} ms | bwd dgrad {t_d:.3f} ms | '
      f'bwd wgrad {t_w:.3f} ms', flush=True)
