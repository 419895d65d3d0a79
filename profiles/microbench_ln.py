"""GPU micro-benchmark of the LayerNorm row kernels (rvt_layernorm_fwd / _bwd) at the four stage shapes of RVT-Base 1 Mpx (bf16):
achieved HBM rate against what torch's own streaming kernels reach on the same box (copy / add)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

dev, dt = torch.device('cuda', 0), torch.bfloat16


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
    return ts[n // 2]


a = torch.empty(1 << 29, dtype=dt, device=dev).normal_()
b = torch.empty_like(a)
gb = a.numel() * 2e-9
print(f'torch copy_: {2 * gb / timeit(lambda: b.copy_(a)):.2f} TB/s   torch add (r2 w1 on halves): '
      f'{3 * gb / 2 / timeit(lambda: torch.add(a[:1 << 28], a[1 << 28:], out=b[:1 << 28])):.2f} TB/s')
del a, b
for rows, C in ((7741440, 64), (1935360, 128), (483840, 256), (120960, 512)):
    x, dy, dres = (torch.randn(rows, C, device=dev).to(dt) for _ in range(3))
    w, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    y = torch.empty_like(x)
    tf = timeit(lambda: ops.layernorm_fwd(x, w, bb, 1e-5, out=y))
    tb = timeit(lambda: ops.layernorm_bwd(x, w, dy, dres, dw, db, 1e-5, out=y))
    tb0 = timeit(lambda: ops.layernorm_bwd(x, w, dy, None, dw, db, 1e-5, out=y))
    g = rows * C * 2e-9
    print(f'rows={rows:8d} C={C:3d}: fwd {tf:.3f} ms = {2 * g / tf:.2f} TB/s | bwd (+dres) {tb:.3f} ms = {4 * g / tb:.2f} TB/s | bwd {tb0:.3f} ms = {3 * g / tb0:.2f} TB/s', flush=True)
    del x, dy, dres, y

# (round 5: a software-pipelined variant - next trip's rows requested before this trip's stores, branch-free buffer accesses - measured
#  0.42 - 0.47 / 0.82 - 0.87 ms at C = 64 against 0.40 / 0.80 for these kernels: eight waves per SIMD already hide the in-order store wait; not kept)
