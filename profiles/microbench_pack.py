"""GPU micro-benchmark of the per-step weight pack (rvt_pack_table over the whole RVT-Base model) and of its parts by descriptor kind."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rvt_amd import _lib as L, weights as Wt

dev = torch.device('cuda', 0)
wl = bench.WORKLOADS['base_1mpx']
m = bench.build_model(wl, torch.bfloat16, dev)
x = bench.make_batch(dict(wl, T=2, B=1), dev, 0)
m.train()
feats, _ = m.forward_sequence(x, None)
mw = m._mw_cache
tab = mw.table


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f'full pack: {timeit(mw.pack):.1f} us, {len(tab)} descriptors, {tab.blocks} blocks')
names = ['COPY', 'TRANSPOSE', 'CONV_FWD', 'CONV_DGRAD', 'LSTM_ROWS', 'CONV_WGRAD_ACC', 'CONV_DGRAD4']
for kind in range(7):
    rows = [r for r in tab.rows if r['kind'] == kind]
    if not rows:
        continue
    sub = Wt._Table(Wt.PACK_DT, 1024)
    for r in rows:
        f = {k: v for k, v in r.items() if k != 'block0'}
        sub.add((int(f['n']) + 1023) // 1024, **f)
    sub.upload(dev)
    t = timeit(lambda: L.call('rvt_pack_table', L.ptr(sub.dev), len(sub), sub.blocks, L.dtype_code(torch.bfloat16), L.stream_of(mw.bufT)))
    print(f'  {names[kind]:<15} {len(rows):4d} descriptors {sum(int(r["n"]) for r in rows) / 1e6:8.2f} M elements {t:8.1f} us')
