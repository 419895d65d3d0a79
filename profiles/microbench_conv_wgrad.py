"""GPU micro-benchmark of the down-sampling conv weight gradients of RVT-Base (stages 2-4, 504 frames at 1 Mpx): the 256-wide
token-contraction kernel with the im2col gather in the LDS-DMA source address (csrc/ppgemm_tn.hpp, CONV) against the split-K
im2col engine (gemm.hpp)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops, tuning

dev, dt = torch.device('cuda', 0), torch.bfloat16


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
    return ts[n // 2]


g = torch.Generator(device=dev).manual_seed(0)
for (Fr, H, W, Cin, Cout) in [(504, 96, 160, 64, 128), (504, 48, 80, 128, 256), (504, 24, 40, 256, 512)]:
    x = torch.randn(Fr, H, W, Cin, device=dev, generator=g).to(dt)
    dy = torch.randn(Fr, H // 2, W // 2, Cout, device=dev, generator=g).to(dt)
    dw = torch.zeros(Cout, 9 * Cin, device=dev)
    gf = 2.0 * Fr * (H // 2) * (W // 2) * Cout * 9 * Cin * 1e-9
    t1 = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 2, 1))
    with tuning.override(conv_wgrad_tn=0):
        t0 = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 2, 1))
    print(f'conv_wgrad {H}x{W} {Cin}->{Cout} ({gf:.0f} GFLOP): default route {t1:.3f} ms = {gf / t1:.0f} TFLOP/s | split-K im2col engine {t0:.3f} ms = {gf / t0:.0f} TFLOP/s', flush=True)
    del x, dy
