"""GPU micro-benchmark: fused MLP kernels vs the op-by-op chain at the RVT-Base 1Mpx stage-1/2 shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

if "--lib" in sys.argv:
    from rvt_amd import _lib
    _lib._install_test_library(_lib.load_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])))
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
    dy = torch.randn(M, C, device=dev, generator=g).to(dt)
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1 = (torch.randn(4 * C, C, device=dev, generator=g) * 0.1).to(dt)
    w2 = (torch.randn(C, 4 * C, device=dev, generator=g) * 0.1).to(dt)
    w2gt, w1t = w2.t().contiguous(), w1.t().contiguous()
    b1, b2, gam = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dlw, dlb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

    t_f = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=True))
    t_fi = timeit(lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=False))

    def chain_f():
        v2 = ops.layernorm_fwd(x, lw, lb, 1e-5)
        h, hp = ops.linear_gelu_fwd(v2, w1, b1, want_grad=True)
        return ops.linear_scale_res_fwd(h, w2, b2, gam, x), h, hp
    t_c = timeit(chain_f)
    _, hg, hgp = chain_f()
    t_b = timeit(lambda: ops.mlp_bwd_dgrad(dy, hgp, x, lw, w2gt, w1t, dlw, dlb, 1e-5))

    def chain_b():
        dhd = ops.linear_dgrad(dy, w2gt, mul=hgp)
        dv2 = ops.linear_dgrad(dhd, w1t)
        return ops.layernorm_bwd(x, lw, dv2, dy, dlw, dlb, 1e-5)
    t_cb = timeit(chain_b)
    print(f'C={C} M={M}: fwd fused(train) {t_f:.3f} ms  fused(infer) {t_fi:.3f} ms  chain {t_c:.3f} ms | '
          f'bwd-dgrad fused {t_b:.3f} ms  chain {t_cb:.3f} ms')
    del x, dy, hg, hgp
