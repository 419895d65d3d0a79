"""GPU micro-benchmark of the linear entry points at the RVT-Base 1Mpx stage-3 / stage-4 shapes (the MFMA-bound ones) and the
per-step ConvLSTM products.  Run once with RVT_PPGEMM=0 (128-row register-staged engine, csrc/gemm.hpp) and once with
RVT_PPGEMM=1 (256 x 256 LDS-DMA ping-pong kernel, csrc/ppgemm.hpp): the switch is read once per process.
Also checks every output against an fp32 torch product on a sample of rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops  # noqa: E402

dev, dt = torch.device('cuda', 0), torch.bfloat16
tag = 'ppgemm=' + os.environ.get('RVT_PPGEMM', '1')
only = [a for a in sys.argv[1:] if not a.startswith('-')]


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
    return ts[n // 2], ts[0]


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dt)


def check(name, got, want_fn, rows):
    want = want_fn(rows).float()
    g = got[rows].float()
    err = (g - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
    if not err < 2e-2:
        print(f'   !! {name}: rel err {err:.3e} on sampled rows')
    return err


def report(name, ms, best, bytes_, flops, err):
    print(f'[{tag}] {name:44s} {ms:7.3f} ms (best {best:7.3f})  {bytes_ / ms / 1e9:5.2f} TB/s  {flops / ms / 1e9:7.1f} TFLOP/s   err {err:.1e}')


SHAPES = [  # (label, M, N, K)
    ('s3 qkv', 483840, 768, 256), ('s3 proj', 483840, 256, 256), ('s3 fc1', 483840, 1024, 256), ('s3 fc2', 483840, 256, 1024),
    ('s4 qkv', 120960, 1536, 512), ('s4 proj', 120960, 512, 512), ('s4 fc1', 120960, 2048, 512), ('s4 fc2', 120960, 512, 2048),
    ('lstm s3 step', 23040, 1024, 512), ('lstm s4 step', 5760, 2048, 1024),
]
for label, M, N, K in SHAPES:
    if only and not any(o in label for o in only):
        continue
    x, w = rnd(M, K), rnd(N, K, scale=0.05)
    b, gam = torch.randn(N, device=dev), torch.randn(N, device=dev)
    res = rnd(M, N)
    y = torch.empty(M, N, device=dev, dtype=dt)
    rows = torch.randint(0, M, (512,), device=dev)
    rows[-1] = M - 1
    ref = lambda r: x[r].float() @ w.float().t()
    fl, by = 2.0 * M * N * K, (M * K + N * K + M * N) * 2
    ms, best = timeit(lambda: ops.linear_fwd(x, w, b, out=y))
    report(f'{label} fwd+bias  {M}x{N}x{K}', ms, best, by, fl, check('fwd', y, lambda r: ref(r) + b, rows))
    ms, best = timeit(lambda: ops.linear_dgrad(x, w, out=y))
    report(f'{label} dgrad plain', ms, best, by, fl, check('dgrad', y, ref, rows))
    ms, best = timeit(lambda: ops.linear_dgrad(x, w, add=res, out=y))
    report(f'{label} dgrad +add', ms, best, by + M * N * 2, fl, check('dgrad add', y, lambda r: ref(r) + res[r].float(), rows))
    ms, best = timeit(lambda: ops.linear_dgrad(x, w, mul=res, out=y))
    report(f'{label} dgrad *mul', ms, best, by + M * N * 2, fl, check('dgrad mul', y, lambda r: ref(r) * res[r].float(), rows))
    ms, best = timeit(lambda: ops.linear_scale_res_fwd(x, w, b, gam, res, out=y))
    report(f'{label} scale_res', ms, best, by + M * N * 2, fl,
           check('scale_res', y, lambda r: res[r].float() + gam * (ref(r) + b), rows))
    # weight gradient dW[N][K] = dy[M][N]^T x[M][K] (+ bias column sums): csrc/ppgemm_tn.hpp vs the 128 x 128 split-K engine
    if M >= 100000:
        dyw, dw, csum = rnd(M, N, scale=0.5), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        ms, best = timeit(lambda: ops.linear_wgrad(dyw, x, dw, colsum_out=csum))
        dw.zero_(); csum.zero_()
        ops.linear_wgrad(dyw, x, dw, colsum_out=csum)
        wr = torch.randint(0, N, (64,), device=dev)
        want = (dyw[:, wr].float().t() @ x.float())
        err = (dw[wr] - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        err2 = (csum - dyw.float().sum(0)).abs().max().item() / max(dyw.float().sum(0).abs().max().item(), 1e-6)
        report(f'{label} wgrad +colsum', ms, best, (M * N + M * K) * 2, fl, max(err, err2))
        del dyw, dw
    if N >= 4 * K:
        g = None
        def run():
            global g
            g = ops.linear_gelu_fwd(x, w, b, want_grad=True)
        ms, best = timeit(run)
        report(f'{label} gelu dual', ms, best, by + M * N * 2, fl,
               check('gelu', g[0], lambda r: torch.nn.functional.gelu(ref(r) + b), rows))
        g = None
    del x, w, res, y
