"""What the box's HBM actually sustains (torch built-in kernels, 4 GiB buffers): read-only, write-only, copy."""
import torch
dev = torch.device('cuda', 0)
n = 1 << 31                      # 2 Gi bf16 elements = 4 GiB
a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
b = torch.empty_like(a)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = a.numel() * 2 / 1e9
t = timeit(lambda: b.fill_(1.0));            print(f'write-only  fill_   : {gb / t:6.2f} TB/s')
t = timeit(lambda: b.copy_(a));              print(f'copy (r+w)  copy_   : {2 * gb / t:6.2f} TB/s total')
t = timeit(lambda: a.view(torch.int16).max()); print(f'read-only   max     : {gb / t:6.2f} TB/s')
t = timeit(lambda: torch.add(a, a, out=b));  print(f'r1 w1       add     : {2 * gb / t:6.2f} TB/s total')
c = torch.empty(n // 4, dtype=torch.bfloat16, device=dev)
t = timeit(lambda: torch.add(a[:n // 4], a[n // 4:n // 2], out=c)); print(f'r2 w1       add     : {3 * gb / 4 / t:6.2f} TB/s total')

# ---- event stream -> stacked histogram (rvt_stacked_histogram): 1 Mpx sensor, 10 bins, 50 ms of events
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd.representations import StackedHistogram
for n in (2_000_000, 20_000_000):
    g = torch.Generator(device='cuda').manual_seed(0)
    H, W, bins = 720, 1280, 10
    ex = torch.randint(0, W, (n,), generator=g, device='cuda'); ey = torch.randint(0, H, (n,), generator=g, device='cuda')
    ep = torch.randint(0, 2, (n,), generator=g, device='cuda')
    et = torch.sort(torch.randint(0, 50_000, (n,), generator=g, device='cuda')).values
    rep = StackedHistogram(bins, H, W, 10, True)
    rep.construct(ex, ey, ep, et); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        rep.construct(ex, ey, ep, et)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    cells = 2 * bins * H * W
    byts = n * 32 + cells * (4 + 4 + 1)          # event records + counter image (memset, read) + uint8 output
    print(f'stacked_histogram n={n}: {ms:.3f} ms  {n / ms / 1e6:.2f} G events/s  {byts / ms / 1e9:.2f} TB/s algorithmic')
