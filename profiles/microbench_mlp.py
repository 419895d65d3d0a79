"""GPU micro-benchmark: fused MLP forward vs the op-by-op chain at the RVT-Base 1Mpx stage shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

dev = torch.device('cuda', 0)
dt = torch.bfloat16


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for C, M in ((64, 7741440), (128, 1935360)):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, C, device=dev, generator=g).to(dt)
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1 = (torch.randn(4 * C, C, device=dev, generator=g) * 0.1).to(dt)
    w2 = (torch.randn(C, 4 * C, device=dev, generator=g) * 0.1).to(dt)
    b1, b2, gam = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    out = torch.empty_like(x)
    t_f = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, out=out))

    def chain():
        v2 = ops.layernorm_fwd(x, lw, lb, 1e-5)
        h, _ = ops.linear_gelu_fwd(v2, w1, b1, want_grad=False)
        return ops.linear_scale_res_fwd(h, w2, b2, gam, x)
    t_c = timeit(chain)
    flops = 2.0 * M * C * 4 * C * 2
    print(f'C={C} M={M}: fused {t_f:.3f} ms ({flops / t_f / 1e9:.0f} TFLOP/s, {2 * M * C * 2 / t_f / 1e9:.2f} TB/s algorithmic), '
          f'chain {t_c:.3f} ms')
