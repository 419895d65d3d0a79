"""GPU micro-benchmark: partition attention core (rvt_attn_fwd / rvt_attn_bwd) at the stage 2-4 shapes of RVT-Base 1Mpx.
Round-1 kernels (transposed LDS copies built with 2-byte stores), measured before their removal: profiles/r2/microbench_attn_core.txt."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

dev, dt = torch.device('cuda', 0), torch.bfloat16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tag = 'old' if os.environ.get('RVT_ATTN_CORE_OLD') == '1' else 'new'
for H, W, C in ((48, 80, 128), (24, 40, 256), (12, 20, 512)):
    F_ = 504
    qkv = torch.randn(F_, H, W, 3 * C, device=dev).to(dt)
    do = torch.randn(F_, H, W, C, device=dev).to(dt)
    for window in (True, False):
        tf = timeit(lambda: ops.attn_fwd(qkv, F_, H, W, C, 32, 6, 10, window))
        tb = timeit(lambda: ops.attn_bwd(qkv, do, F_, H, W, C, 32, 6, 10, window))
        print(f'[{tag}] {H}x{W} C={C} window={int(window)}: fwd {tf * 1e3:7.1f} us  bwd {tb * 1e3:7.1f} us', flush=True)
