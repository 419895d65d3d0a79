"""GPU micro-benchmark: stage-1 ConvLSTM reverse scan (368640 pixels x 21 steps, C = 64, bf16, in-kernel weight gradients)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops
torch.manual_seed(0)
dev, dt = torch.device('cuda', 0), torch.bfloat16
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
T_, Mp, Cc = 21, 368640, 64
xa, Hall, Cs = rnd(T_, Mp, Cc), rnd(T_ + 1, Mp, Cc) * 0.5, rnd(T_, Mp, Cc)
w, b = rnd(4 * Cc, 2 * Cc) * 0.1, torch.zeros(4 * Cc, device=dev)
dH, dxa = rnd(T_, Mp, Cc), torch.empty(T_, Mp, Cc, device=dev, dtype=dt)
dh0, dc0 = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev)
wt = w.t().contiguous()
def run():
    dw, db = torch.zeros(4 * Cc, 2 * Cc, device=dev), torch.zeros(4 * Cc, device=dev)
    ops.lstm_scan_bwd(xa, Hall, Cs, None, dH, None, w, wt, b, dxa, None, dh0, dc0, dw=dw, db=db)
    return dw, db
dw, db = run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print(f'[HELP={os.environ.get("RVT_LSTM_SCAN_HELP", "0")}] lstm_scan_bwd C=64: {e0.elapsed_time(e1) / 5:.3f} ms   checksum dw {float(dw.double().abs().sum()):.6e} db {float(db.double().abs().sum()):.6e} dx {float(dxa.double().abs().sum()):.6e}', flush=True)
