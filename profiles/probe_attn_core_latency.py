"""Is the partition-attention core (stage 2 of RVT-Base: 48x80, C=128) bound by memory latency?  The same kernel on 504 frames
(3.5 GB per backward launch, HBM) and on 16 frames looped (112 MB: resident in the 256 MB MALL / L2): time per frame."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops
dev, dt = torch.device('cuda', 0), torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


H, W, C = 48, 80, 128
for F_ in (504, 126, 32, 16):
    qkv = torch.randn(F_, H, W, 3 * C, device=dev).to(dt)
    do = torch.randn(F_, H, W, C, device=dev).to(dt)
    for window in (True, False):
        tf = timeit(lambda: ops.attn_fwd(qkv, F_, H, W, C, 32, 6, 10, window))
        tb = timeit(lambda: ops.attn_bwd(qkv, do, F_, H, W, C, 32, 6, 10, window))
        print(f'F={F_:4d} window={int(window)}: fwd {tf * 1e3:7.1f} us = {tf * 1e3 / F_:6.3f} us/frame   bwd {tb * 1e3:7.1f} us = {tb * 1e3 / F_:6.3f} us/frame', flush=True)
