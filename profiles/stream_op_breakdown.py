"""Per-entry-point HIP-event time of ONE streaming-inference step (RVT-Base, B = 64, T = 1, bf16; BASELINE configs[4])."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench, opmodel
from rvt_amd import tuning
tuning.use(route_stage_driver=0)          # the Python host loop: every launch goes through _lib.call and can be timed
dev, dt = torch.device('cuda', 0), torch.bfloat16
wl = dict(bench.WORKLOADS['base_1mpx'])
model = bench.build_model(wl, dt, dev)
g = torch.Generator(device=dev).manual_seed(3)
frames = [torch.randint(0, 11, (64, 20, *wl['hw']), generator=g, dtype=torch.uint8, device=dev) for _ in range(2)]
states = None
with torch.no_grad():
    for i in range(6):
        _, states = model(frames[i % 2], states)
    tm = bench.OpTimer(); tm.install()
    _, states = model(frames[0], states)
    tm.uninstall(); torch.cuda.synchronize()
rows = []
for name, recs in tm.records.items():
    for a, b, args in recs:
        rows.append((a.elapsed_time(b), name, bench.shape_key(args)))
tot = sum(r[0] for r in rows)
print(f'# one streaming step, {len(rows)} launches, sum of HIP-event times {tot:.3f} ms (stage driver: events are recorded around the C call of a whole stage)')
by = {}
for ms, n, k in rows:
    e = by.setdefault((n, k), [0, 0.0]); e[0] += 1; e[1] += ms
for (n, k), (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f'{n:28s} {str(k):56s} calls={c:3d} {ms:8.3f} ms')
