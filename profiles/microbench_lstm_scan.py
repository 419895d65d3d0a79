"""GPU micro-benchmark of the ConvLSTM scan kernels at the RVT-Base 1Mpx stage-1 / stage-2 shapes (T = 21, B = 24):
lstm_scan.hpp (round 2/3, N-form) against lstm_scan2.hpp (round 4, T-form) through tuning.lstm_scan_v2, each checked
against the other (same arithmetic, different summation order)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops, tuning

dev, dt, T = torch.device('cuda', 0), torch.bfloat16, 21


def timeit(fn, n=9):
    """median of n single launches (boxes differ by up to 20 %: compare variants inside ONE run, against lstm_scan_v2=0)"""
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


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-9))


for C, M in ((64, 368640), (128, 92160)):
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
    x, h0, c0 = rn(T, M, C).to(dt), rn(M, C, sc=0.5).to(dt), rn(M, C, sc=0.7)
    w, b = rn(4 * C, 2 * C, sc=2.0 / (2 * C) ** 0.5).to(dt), rn(4 * C, sc=0.2)
    wt = w.t().contiguous()
    dH, dcl = rn(T, M, C).to(dt), rn(M, C)
    res = {}
    for v2 in ((0, 1) if C == 64 else (0,)):
        with tuning.override(lstm_scan_v2=v2, route_lstm_scan=1):
            Hall = torch.empty(T + 1, M, C, dtype=dt, device=dev); Hall[0].copy_(h0)
            c_last = torch.empty(M, C, device=dev)
            Csave = torch.empty(T, M, C, dtype=dt, device=dev)
            gsave = torch.empty(T, M, 4 * C, dtype=dt, device=dev) if ops.lstm_scan_saves_gates(dt, C) else None
            t_f = timeit(lambda: ops.lstm_scan_fwd(x, Hall, c0, c_last, Csave, w, b, gates_out=gsave))
            dx, dh0, dc0 = torch.empty(T, M, C, dtype=dt, device=dev), torch.empty(M, C, dtype=dt, device=dev), torch.empty(M, C, device=dev)
            wg = ops.lstm_scan_wgrad_supported(dt, C, M) and gsave is None
            dw, db = torch.zeros(4 * C, 2 * C, device=dev), torch.zeros(4 * C, device=dev)
            dz = None if wg else torch.empty(T, M, 4 * C, dtype=dt, device=dev)
            run = lambda: ops.lstm_scan_bwd(x, Hall, Csave, c0, dH, dcl, w, wt, b, dx, dz, dh0, dc0, dw=dw if wg else None,
                                            db=db if wg else None, gates=gsave)
            t_b = timeit(run)
            dw.zero_(); db.zero_(); run()
            res[v2] = (Hall.clone(), c_last.clone(), dx.clone(), dh0.clone(), dc0.clone(), dw.clone(), db.clone())
            print(f'C={C} M={M} T={T} lstm_scan_v2={v2}: fwd {t_f:.3f} ms | bwd{" (+ in-kernel wgrad)" if wg else ""} {t_b:.3f} ms', flush=True)
    if len(res) == 2:
        names = ('Hall', 'c_last', 'dx', 'dh0', 'dc0', 'dw', 'db')
        print('   v2 vs v1 rel err: ' + ', '.join(f'{n} {rel(a, b_):.2e}' for n, a, b_ in zip(names, res[1], res[0])), flush=True)
