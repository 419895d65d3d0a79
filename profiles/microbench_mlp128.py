"""GPU micro-benchmark of the stage-2 (C = 128, 1.94 M tokens) MLP half on the op-by-op backward route:
fused forward saving GELU + GELU' (round 3) against the forward that saves the pre-activation only (round 4) with GELU
applied on load in the fc2 weight gradient and GELU' in the epilogue of the fc2 input gradient."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

dev, dt = torch.device('cuda', 0), torch.bfloat16
C, M = 128, 1935360


def timeit(fn, n=9):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[n // 2]


g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
x, dy = rn(M, C).to(dt), rn(M, C).to(dt)
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
w1, w2 = rn(4 * C, C, sc=0.1).to(dt), rn(C, 4 * C, sc=0.1).to(dt)
b1, b2, gam = rn(4 * C, sc=0.1), rn(C, sc=0.1), torch.ones(C, device=dev)
w2t = w2.t().contiguous()
s2, cs2 = torch.zeros(C, 4 * C, device=dev), torch.zeros(C, device=dev)
t_f_dual = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=True, want_v2=True))
t_f_pre = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_pre=True, want_v2=True))
t_f_inf = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5))
_, hg, hgp, _ = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=True, want_v2=True)
_, hpre, _, _ = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_pre=True, want_v2=True)
t_w_dual = timeit(lambda: ops.linear_wgrad(dy, hg, s2, colsum_out=cs2))
t_w_pre = timeit(lambda: ops.linear_wgrad(dy, hpre, s2, gelu_in=True, colsum_out=cs2))
t_d_dual = timeit(lambda: ops.linear_dgrad(dy, w2t, mul=hgp))
t_d_pre = timeit(lambda: ops.linear_dgrad(dy, w2t, gelu_pre=hpre))
print(f'C={C} M={M}: forward dual {t_f_dual:.3f} | pre {t_f_pre:.3f} | nothing saved {t_f_inf:.3f} ms')
print(f'   fc2 wgrad: from GELU(h) {t_w_dual:.3f} | GELU on load {t_w_pre:.3f} ms;  fc2 dgrad: * GELU\'(saved) {t_d_dual:.3f} | GELU\' in the epilogue {t_d_pre:.3f} ms')
print(f'   MLP half per block (these three): dual {t_f_dual + t_w_dual + t_d_dual:.3f} ms | pre {t_f_pre + t_w_pre + t_d_pre:.3f} ms', flush=True)
