"""Phase timers of lstm_scan2_bwd_kernel (measurement build with -DSCAN2_PROF, loaded through RVT_HIP_LIB): shader-clock cycles
wave 0 of workgroup 0 spends in each phase of a step, averaged over its steps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops
dev, dt = torch.device('cuda', 0), torch.bfloat16
T_, Mp, Cc = 21, 368640, 64
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
xa, Hall, Cs = rnd(T_, Mp, Cc), rnd(T_ + 1, Mp, Cc) * 0.5, rnd(T_, Mp, Cc)
wl, bl = rnd(4 * Cc, 2 * Cc) * 0.1, torch.zeros(4 * Cc, device=dev)
dH, dxa = rnd(T_, Mp, Cc), torch.empty(T_, Mp, Cc, device=dev, dtype=dt)
dh0, dc0 = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev)
wt = wl.t().contiguous()
dw, db = torch.zeros(4 * Cc, 2 * Cc, device=dev), torch.zeros(4 * Cc, device=dev)
fn = lambda: ops.lstm_scan_bwd(xa, Hall, Cs, None, dH, None, wl, wt, bl, dxa, None, dh0, dc0, dw=dw, db=db)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
ws = ops._WS[('scanbwd', 'cuda', 0, int(torch.cuda.current_stream().cuda_stream))]
grid = 256
rec = 4 * Cc * 2 * Cc + 4 * Cc
t = ws[grid * rec: grid * rec + 8].cpu().tolist()
tiles = (Mp // 64 + grid - 1) // grid
steps = tiles * T_
names = ['P2', 'P1 + park', 'dx epilogue', 'gate backward', 'barrier B wait', 'write dz + fetch', 'barrier A wait']
print(f'launch {e0.elapsed_time(e1):.3f} ms; wave 0 of workgroup 0: {tiles} tiles x {T_} steps; s_memtime ticks per step:')
for n, v in zip(names, t):
    print(f'  {n:20s} {v / steps:9.1f}')
print(f'  {"sum":20s} {sum(t[:7]) / steps:9.1f}')
