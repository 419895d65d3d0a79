"""GPU micro-benchmark of norm1 + qkv at C = 128 (stage 2 of RVT-Base: 1.94 M tokens): csrc/ln_linear.hpp (one launch, weights
resident in LDS, rows in registers) against rvt_layernorm_fwd + rvt_linear_fwd."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops, tuning

dev, dt = torch.device('cuda', 0), torch.bfloat16
C, N = 128, 384
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1935360


def timeit(fn, n=15):
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
x = (torch.randn(M, C, device=dev, generator=g) * 1.5).to(dt)
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
w, b = (torch.randn(N, C, device=dev, generator=g) * 0.1).to(dt), torch.randn(N, device=dev, generator=g) * 0.1
u = torch.empty_like(x)
y = torch.empty(M, N, device=dev, dtype=dt)
t_ln = timeit(lambda: ops.layernorm_fwd(x, lw, lb, 1e-5, out=u))
t_lin = timeit(lambda: ops.linear_fwd(u, w, b, out=y))
t_f = timeit(lambda: ops.ln_linear_fwd(x, lw, lb, w, b, 1e-5, want_u=True))
t_fn = timeit(lambda: ops.ln_linear_fwd(x, lw, lb, w, b, 1e-5, want_u=False))
gb = lambda rows: rows * M * C * 2e-9
print(f'C={C} N={N} M={M}: layernorm_fwd {t_ln[0]:.3f} + linear_fwd {t_lin[0]:.3f} = {t_ln[0] + t_lin[0]:.3f} ms | ln_linear (u kept) '
      f'{t_f[0]:.3f} (min {t_f[1]:.3f}) ms = {gb(5) / t_f[0]:.2f} TB/s of rows | ln_linear (no u) {t_fn[0]:.3f} ms = {gb(4) / t_fn[0]:.2f} TB/s')
for res in (128, 512):
    with tuning.override(chain_resident=res):
        t = timeit(lambda: ops.ln_linear_fwd(x, lw, lb, w, b, 1e-5, want_u=True))
    print(f'   chain_resident={res}: {t[0]:.3f} ms')

# fc1 + GELU (+ GELU') at K = 256, N = 1024 (stage 3 of RVT-Base: 484 k tokens): weight-stationary column groups against the GEMM engine
C2, N2, M2 = 256, 1024, 483840
u2 = (torch.randn(M2, C2, device=dev, generator=g)).to(dt)
w2, b2 = (torch.randn(N2, C2, device=dev, generator=g) * 0.1).to(dt), torch.randn(N2, device=dev, generator=g) * 0.1
t_new = timeit(lambda: ops.linear_gelu_fwd(u2, w2, b2, want_grad=True))
t_newi = timeit(lambda: ops.linear_gelu_fwd(u2, w2, b2, want_grad=False))
with tuning.override(ln_linear=0):
    t_old = timeit(lambda: ops.linear_gelu_fwd(u2, w2, b2, want_grad=True))
    t_oldi = timeit(lambda: ops.linear_gelu_fwd(u2, w2, b2, want_grad=False))
gb2 = (M2 * C2 + 2 * M2 * N2) * 2e-9
print(f'linear_gelu_fwd K={C2} N={N2} M={M2}: weight-stationary (g, gp) {t_new[0]:.3f} (min {t_new[1]:.3f}) ms = {gb2 / t_new[0]:.2f} TB/s of rows, '
      f'g only {t_newi[0]:.3f} ms | GEMM engine {t_old[0]:.3f} ms, g only {t_oldi[0]:.3f} ms')
