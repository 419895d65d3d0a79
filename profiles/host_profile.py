"""Host-side (Python) profile of the training step: where the enqueue time goes.  usage: python profiles/host_profile.py"""
import cProfile, pstats, sys, os, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench

wl = dict(bench.WORKLOADS[os.environ.get('HP_WL', 'base_1mpx')]); wl['B'] = int(os.environ.get('HP_B', wl['B']))
dev = torch.device('cuda', 0)
model = bench.build_model(wl, torch.bfloat16, dev)
params = list(model.parameters())
opt = torch.optim.AdamW(params, lr=2e-4, fused=True)
xs = bench.make_batch(wl, dev, 1)
T, B = wl['T'], wl['B']
geoms = model.stage_geoms(*model.in_res_hw)
cots = {s + 1: torch.randn((T, B, geoms[s].H, geoms[s].W, geoms[s].C), device=dev, dtype=torch.bfloat16).permute(0, 1, 4, 2, 3) for s in (1, 2, 3)}


def step():
    feats, states = model.forward_sequence(xs, None)
    torch.autograd.backward([feats[s] for s in (2, 3, 4)], [cots[s] for s in (2, 3, 4)])
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    step()
host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
print(f'host enqueue per step: {host * 1e3:.1f} ms')
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue())
