"""GPU micro-benchmark of the GEMM-family entry points at RVT-Base 1Mpx stage-1/2 shapes: achieved TB/s and TFLOP/s."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

dev = torch.device('cuda', 0)
dt = torch.bfloat16


if '--lib' in sys.argv:       # A/B a differently-built library (tuning only)
    from rvt_amd import _lib
    _lib._install_test_library(_lib.load_library(os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])))


def timeit(fn, n=9):
    """median of n individually timed launches after 3 warm-ups"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]


def rnd(*shape):
    return torch.randn(*shape, device=dev).to(dt)


def report(name, ms, bytes_, flops):
    print(f'{name:46s} {ms:7.3f} ms  {bytes_ / ms / 1e9:6.2f} TB/s  {flops / ms / 1e9:7.1f} TFLOP/s')


for C, M in ((64, 7741440), (128, 1935360)):
    x, dy4, w1 = rnd(M, C), rnd(M, 4 * C), rnd(4 * C, C)
    b4 = torch.zeros(4 * C, device=dev)
    y4 = torch.empty(M, 4 * C, device=dev, dtype=dt)
    ms = timeit(lambda: ops.linear_fwd(x, w1, b4, out=y4))
    report(f'linear_fwd  M={M} K={C} N={4 * C}', ms, M * C * 2 + M * 4 * C * 2, 2.0 * M * C * 4 * C)
    ms = timeit(lambda: ops.linear_gelu_fwd(x, w1, b4, want_grad=True))
    report(f'linear_gelu_fwd (g+gp)  K={C} N={4 * C}', ms, M * C * 2 + 2 * M * 4 * C * 2, 2.0 * M * C * 4 * C)
    wt = rnd(C, 4 * C)
    dxo = torch.empty(M, C, device=dev, dtype=dt)
    ms = timeit(lambda: ops.linear_dgrad(dy4, wt, out=dxo))
    report(f'linear_dgrad N={4 * C} -> K={C}', ms, M * 5 * C * 2, 2.0 * M * C * 4 * C)
    wt2 = rnd(4 * C, C)
    ms = timeit(lambda: ops.linear_dgrad(x, wt2, mul=dy4, out=y4))
    report(f'linear_dgrad N={C} -> K={4 * C} (*mul)', ms, M * 9 * C * 2, 2.0 * M * C * 4 * C)
    dw = torch.zeros(4 * C, C, device=dev)
    cs = torch.zeros(4 * C, device=dev)
    ms = timeit(lambda: ops.linear_wgrad(dy4, x, dw, colsum_out=cs))
    report(f'linear_wgrad dy[{4 * C}] x[{C}] +colsum', ms, M * 5 * C * 2, 2.0 * M * C * 4 * C)
    ms = timeit(lambda: ops.linear_wgrad(dy4, x, dw))
    report(f'linear_wgrad dy[{4 * C}] x[{C}]', ms, M * 5 * C * 2, 2.0 * M * C * 4 * C)
    dw2 = torch.zeros(C, 4 * C, device=dev)
    cs2 = torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.linear_wgrad(x, dy4, dw2, colsum_out=cs2))
    report(f'linear_wgrad dy[{C}] x[{4 * C}] +colsum', ms, M * 5 * C * 2, 2.0 * M * C * 4 * C)
    dw3 = torch.zeros(C, C, device=dev)
    x2 = rnd(M, C)
    ms = timeit(lambda: ops.linear_wgrad(x, x2, dw3, colsum_out=cs2))
    report(f'linear_wgrad dy[{C}] x[{C}] +colsum', ms, M * 2 * C * 2, 2.0 * M * C * C)
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.layernorm_fwd(x, lw, lb, 1e-5, out=dxo))
    report(f'layernorm_fwd C={C}', ms, M * 2 * C * 2, 0)
    dyc = rnd(M, C)
    dlw, dlb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.layernorm_bwd(x, lw, dyc, None, dlw, dlb, 1e-5))
    report(f'layernorm_bwd C={C}', ms, M * 3 * C * 2, 0)
    ms = timeit(lambda: ops.layernorm_bwd(x, lw, dyc, x2, dlw, dlb, 1e-5))
    report(f'layernorm_bwd C={C} +dres', ms, M * 4 * C * 2, 0)
    del x, dy4, y4, dxo, x2, dyc

# write-pattern probe: same K, output width = 64 / 128 (full rows per tile) vs 256 (two tiles per row)
M, C = 7741440, 64
x = rnd(M, C)
for N in (64, 128, 256):
    w = rnd(N, C)
    b = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=dt)
    ms = timeit(lambda: ops.linear_fwd(x, w, b, out=y))
    report(f'probe linear_fwd K=64 N={N}', ms, M * C * 2 + M * N * 2, 2.0 * M * C * N)
    del y
