"""Times rvt_lstm_dgrad at the RVT-Tiny stage-4 per-step shape (M = 640, C = 256) in isolation: HIP events and host wall clock."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops
dev, dt = torch.device('cuda', 0), torch.bfloat16
for M, C in ((640, 256), (5760, 512), (2560, 128)):
    dz = torch.randn(M, 4 * C, device=dev).to(dt)
    wt = torch.randn(2 * C, 4 * C, device=dev).to(dt)
    dx, dh = torch.empty(M, C, device=dev, dtype=dt), torch.empty(M, C, device=dev, dtype=dt)
    for _ in range(3):
        ops.lstm_dgrad(dz, wt, dx, dh)
    torch.cuda.synchronize()
    ev, wall = [], []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); ops.lstm_dgrad(dz, wt, dx, dh); e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize()
        ev.append(e0.elapsed_time(e1)); wall.append((t1 - t0) * 1e3)
    ev.sort(); wall.sort()
    print(f'lstm_dgrad M={M} C={C}: events median {ev[10]:.4f} ms (min {ev[0]:.4f}, max {ev[-1]:.4f}); host call median {wall[10]:.4f} ms (max {wall[-1]:.4f})')
