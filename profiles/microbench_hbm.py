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
