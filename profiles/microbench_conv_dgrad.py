"""GPU micro-benchmark: input gradient of the stage 2-4 down-sampling convs (3x3 / 2 / 1) at the RVT-Base 1Mpx shapes -
four parity-class GEMM launches (rvt_conv_dgrad) against the one-launch 2x2-block product (rvt_conv_dgrad4, csrc/ppgemm.hpp GATHER)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops, weights

dev, dt = torch.device('cuda', 0), torch.bfloat16


def timeit(fn, n=9):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]


F_ = 504
for stage, (H, W, Cin, Cout) in ((2, (96, 160, 64, 128)), (3, (48, 80, 128, 256)), (4, (24, 40, 256, 512))):
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05)
    dy = torch.randn(F_, H // 2, W // 2, Cout, device=dev).to(dt)
    add = torch.randn(F_, H, W, Cin, device=dev).to(dt)
    wd, wd4 = weights.pack_conv_dgrad(w, 2, 1, dt), weights.pack_conv_dgrad4(w, dt)
    fl = 2.0 * F_ * (H // 2) * (W // 2) * 9 * Cin * Cout
    for a, tag in ((None, ''), (add, ' + add')):
        t_old = timeit(lambda: ops.conv_dgrad(dy, wd, a, H, W, Cin, 3, 2, 1))
        t_new = timeit(lambda: ops.conv_dgrad4(dy, wd4, a, H, W, Cin))
        o, n = ops.conv_dgrad(dy, wd, a, H, W, Cin, 3, 2, 1), ops.conv_dgrad4(dy, wd4, a, H, W, Cin)
        err = float((o.float() - n.float()).abs().max() / o.float().abs().max())
        print(f'stage {stage} conv dgrad{tag:6s} ({F_}x{H}x{W}x{Cin} <- {Cout}): parity-class launches {t_old:.3f} ms ({fl / t_old / 1e9:.0f} TFLOP/s useful) | '
              f'one launch {t_new:.3f} ms ({fl / t_new / 1e9:.0f} TFLOP/s useful)   max diff / scale {err:.1e}', flush=True)
