"""Wide-stage ConvLSTM: one scan launch (csrc/lstm_scan3.hpp) against the per-step kernels it replaces, at the shapes of
RVT-Base stage 3 (M = 23040 tokens per step, C = 256, T = 21) and RVT-Tiny stage 4 (M = 640)."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from rvt_amd import ops, tuning
dev, dt = torch.device('cuda', 0), torch.bfloat16
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
T = 21
for M, C in ((23040, 256), (640, 256), (92160, 128), (2560, 128)):
    x = rnd(T, M, C)
    w = rnd(4 * C, 2 * C) * 0.05
    b = torch.zeros(4 * C, device=dev)
    wp, wtp = ops.lstm_scan3_pack(w)
    dH, dc_last = rnd(T, M, C), torch.randn(M, C, device=dev)
    dx, dz = torch.empty(T, M, C, device=dev, dtype=dt), torch.empty(T, M, 4 * C, device=dev, dtype=dt)
    dh0, dc0 = torch.empty(M, C, device=dev, dtype=dt), torch.empty(M, C, device=dev)
    c_last = torch.empty(M, C, device=dev)
    for rb in (1, 2):
        with tuning.override(lstm_scan3_rb256=rb, lstm_scan3_rb128=rb):
            rows = ops.lstm_scan3_rows(C, M)
            Hall = torch.zeros(T + 1, M, C, device=dev, dtype=dt)
            Cs, gs = torch.empty(T, rows, C, device=dev, dtype=dt), torch.empty(T, rows, 4 * C, device=dev, dtype=dt)
            tf = timeit(lambda: ops.lstm_scan3_fwd(x, Hall, None, c_last, Cs, wp, b, gs))
            tn = timeit(lambda: ops.lstm_scan3_fwd(x, Hall, None, c_last, None, wp, b, None))
            tb = timeit(lambda: ops.lstm_scan3_bwd(gs, Cs, None, dH, dc_last, wtp, dx, dz, dh0, dc0))
            print(f'M={M} C={C} rb={rb}: scan3 fwd {tf:.3f} ms (no-grad {tn:.3f})  bwd {tb:.3f} ms', flush=True)
    tp = timeit(lambda: ops.lstm_scan3_pack(w))
    # the per-step kernels
    xs, hs, cs_ = rnd(M, C), rnd(M, C), torch.randn(M, C, device=dev)
    perm_w = rnd(4 * C, 2 * C) * 0.05
    ho, co, go = torch.empty(M, C, device=dev, dtype=dt), torch.empty(M, C, device=dev), torch.empty(M, 4 * C, device=dev, dtype=dt)
    dzs, wt = rnd(M, 4 * C), rnd(2 * C, 4 * C) * 0.05
    dxs, dhs = torch.empty(M, C, device=dev, dtype=dt), torch.empty(M, C, device=dev, dtype=dt)
    dhin, dcr = rnd(M, C), torch.randn(M, C, device=dev)
    t1 = timeit(lambda: ops.lstm_fwd(xs, hs, cs_, perm_w, b, ho, co, go))
    t2 = timeit(lambda: ops.lstm_dgrad(dzs, wt, dxs, dhs))
    t3 = timeit(lambda: ops.lstm_gates_bwd(dhin, dhs, dcr, go, co, cs_, dzs))
    print(f'M={M} C={C}: pack {tp * 1e3:.1f} us;  per-step route x{T}: fwd {T * t1:.3f} ms, bwd {T * (t2 + t3):.3f} ms', flush=True)
    if C == 128:        # the register-resident-weight scan of lstm_scan.hpp (what stage 2 of RVT-Base ran before)
        Hall = torch.zeros(T + 1, M, C, device=dev, dtype=dt)
        Cs, gs = torch.empty(T, M, C, device=dev, dtype=dt), torch.empty(T, M, 4 * C, device=dev, dtype=dt)
        tf = timeit(lambda: ops.lstm_scan_fwd(x, Hall, None, c_last, Cs, w, b, gates_out=gs))
        tb = timeit(lambda: ops.lstm_scan_bwd(x, Hall, Cs, None, dH, dc_last, w, w.t().contiguous(), b, dx, dz, dh0, dc0, gates=gs))
        print(f'M={M} C={C}: lstm_scan.hpp (weights in registers, saved gates): fwd {tf:.3f} ms  bwd {tb:.3f} ms', flush=True)
