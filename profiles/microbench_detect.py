"""GPU micro-benchmark of the detection tail on RVT-Base / 1 Mpx shapes (SURVEY.md section 8 rows f2 + f3): YOLOX PAFPN + YOLOX head
forward + backward on N labelled frames of 384x640 (stage 2-4 features 48x80x128, 24x40x256, 12x20x512; A = 5040 anchors),
and the SimOTA / loss tail on its own.  Usage: python profiles/microbench_detect.py [N] [G]"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import fpn as F_, head as H_

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev, dt = torch.device('cuda', 0), torch.bfloat16
chans, hws, strides, nc = (128, 256, 512), ((48, 80), (24, 40), (12, 20)), (8, 16, 32), 3


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
    return ts[n // 2]


torch.manual_seed(0)
fpn = F_.YOLOPAFPN(depth=0.67, in_channels=chans, compute_dtype=dt).to(dev).train()
head = H_.YOLOXHead(num_classes=nc, strides=strides, in_channels=chans, compute_dtype=dt).to(dev).train()
g = torch.Generator(device=dev).manual_seed(1)
feats = {s: torch.randn(N, h, w, c, device=dev, generator=g).to(dt).permute(0, 3, 1, 2).requires_grad_(True)
         for s, c, (h, w) in zip((2, 3, 4), chans, hws)}
labels = torch.zeros(N, G, 5, device=dev)
for b in range(N):
    n = int(torch.randint(0, G + 1, (1,), generator=torch.Generator().manual_seed(b)))
    r = torch.rand(n, 5, generator=torch.Generator().manual_seed(100 + b))
    labels[b, :n] = torch.stack([(r[:, 0] * nc).floor(), 20 + r[:, 1] * 600, 20 + r[:, 2] * 344, 16 + r[:, 3] * 200, 16 + r[:, 4] * 150], 1).to(dev)


def zero():
    fpn.zero_grad(set_to_none=True); head.zero_grad(set_to_none=True)
    for f in feats.values():
        f.grad = None


def step():
    zero()                                            # (as a training loop does: without it autograd adds into 200+ existing .grad tensors)
    outs = fpn(feats)
    det, losses = head(outs, labels)
    losses['loss'].backward()


def fpn_only():
    zero()
    outs = fpn(feats)
    sum(o.float().sum() for o in outs).backward()


t_all = timeit(step)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t_all_host = (time.perf_counter() - t0) / 5 * 1e3        # enqueue time into an (almost) empty queue
torch.cuda.synchronize()
t_fpn = timeit(fpn_only)
with torch.no_grad():
    outs = fpn(feats)
maps, hw = head._pred_maps([o.detach() for o in outs])
maps = [m.detach().requires_grad_(True) for m in maps]


def tail():
    det, ls, _, _ = H_.simota_loss(maps, labels, hw, strides, nc)
    ls[0].backward()


t_tail = timeit(tail)
t0 = time.perf_counter()
for _ in range(20):
    tail()
t_host = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
fpn.eval(); head.eval()
with torch.no_grad():
    t_inf = timeit(lambda: head(fpn(feats)))
fpn.train(); head.train()
step()
# per-entry-point roofline of ONE training step of the tail (HIP events around every C-ABI launch; opmodel.py prices them)
import bench, opmodel
tm = bench.OpTimer(); tm.install()
fpn.train(); head.train()
step()
tm.uninstall(); torch.cuda.synchronize()
groups = {}
for name, recs in tm.records.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in recs)
    fb = [opmodel.model(name, r[2]) for r in recs]
    ent = dict(kernel=name, launches=len(recs), ms=round(ms, 3))
    if all(x is not None for x in fb):
        ent.update({k: v for k, v in opmodel.roofline_entry(name, sum(x[0] for x in fb), sum(x[1] for x in fb), ms, len(recs), 'bf16').items()
                    if k in ('bound', 'frac', 'hbm_gbs', 'mfma_tflops', 'algorithmic_gbyte', 'algorithmic_gflop')})
    groups[name] = ent
table = sorted(groups.values(), key=lambda e: -e['ms'])
tot = sum(e['ms'] for e in table)
print(f'# detection tail, one training step (N={N} frames, bf16): {len(table)} entry points, {sum(e["launches"] for e in table)} launches, '
      f'sum of kernel time {tot:.2f} ms')
for e in table:
    print('#  ' + json.dumps(e))
fpn.eval(); head.eval()
print(json.dumps({'N': N, 'G': G, 'anchors': 5040, 'dtype': 'bf16', 'train_fpn_head_loss_fwd_bwd_ms': round(t_all, 3),
                  'train_fpn_head_loss_host_enqueue_ms': round(t_all_host, 3), 'train_fpn_fwd_bwd_ms': round(t_fpn, 3), 'decode_simota_loss_fwd_bwd_ms': round(t_tail, 3),
                  'decode_simota_loss_host_enqueue_ms': round(t_host, 3), 'eval_fpn_head_decode_ms': round(t_inf, 3),
                  'num_fg': int((head.last_match >= 0).sum()) if hasattr(head, 'last_match') else None}), flush=True)
