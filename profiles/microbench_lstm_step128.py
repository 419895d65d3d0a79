import sys, os, torch
sys.path.insert(0, os.getcwd())
from rvt_amd import ops
dev, dt = torch.device('cuda', 0), torch.bfloat16
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
for Mp, Cc in ((92160, 128), (23040, 256), (5760, 512)):
    xs, hs, cs_ = rnd(Mp, Cc), rnd(Mp, Cc), torch.randn(Mp, Cc, device=dev)
    wl, bl = rnd(4 * Cc, 2 * Cc) * 0.05, torch.zeros(4 * Cc, device=dev)
    ho, co, go = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev), torch.empty(Mp, 4 * Cc, device=dev, dtype=dt)
    dz, wt = rnd(Mp, 4 * Cc), rnd(2 * Cc, 4 * Cc) * 0.05
    dx, dh = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev, dtype=dt)
    dhin, dcr = rnd(Mp, Cc), torch.randn(Mp, Cc, device=dev)
    tf = timeit(lambda: ops.lstm_fwd(xs, hs, cs_, wl, bl, ho, co, go))
    td = timeit(lambda: ops.lstm_dgrad(dz, wt, dx, dh))
    tg = timeit(lambda: ops.lstm_gates_bwd(dhin, dh, dcr, go, co, cs_, dz))
    print(f'M={Mp} C={Cc}: lstm_fwd {tf*1e3:.1f} us  gates_bwd {tg*1e3:.1f} us  lstm_dgrad {td*1e3:.1f} us  -> T=21: fwd {21*tf:.2f} ms, bwd {21*(tg+td):.2f} ms', flush=True)
