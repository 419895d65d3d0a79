"""End-to-end DETECTOR training step on one MI355X (not the BASELINE metric — that is the backbone step of bench.py): RVT-Base backbone
over the whole (T, B) sequence, the K labelled frames gathered on the device (rvt_gather_frames, modules/utils/detection.py:32-46),
YOLOX PAFPN + head + SimOTA losses on them, backward through everything, fused AdamW over all parameters.  Synthetic data, random-init
weights.  Usage: python profiles/bench_detector_step.py [K labelled frames, default 96] [steps, default 10]"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
from rvt_amd import fpn as F_, head as H_, ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 96
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev, dt = torch.device('cuda', 0), torch.bfloat16
wl = dict(bench.WORKLOADS['base_1mpx'])
T, B = wl['T'], wl['B']
model = bench.build_model(wl, dt, dev)
dims, strides = model.get_stage_dims((2, 3, 4)), model.get_strides((2, 3, 4))
neck = F_.YOLOPAFPN(depth=0.67, in_channels=dims, compute_dtype=dt).to(dev).train()
head = H_.YOLOXHead(num_classes=3, strides=strides, in_channels=dims, compute_dtype=dt).to(dev).train()
params = list(model.parameters()) + list(neck.parameters()) + list(head.parameters())
opt = torch.optim.AdamW(params, lr=2e-4, fused=True)
xs = bench.make_batch(wl, dev, seed=1)
idx = torch.linspace(0, T * B - 1, K, device=dev).round().to(torch.int32)          # which (t, b) frames carry labels
g = torch.Generator().manual_seed(0)
G = 16
labels = torch.zeros(K, G, 5)
for b in range(K):
    n = int(torch.randint(1, G + 1, (1,), generator=g))
    r = torch.rand(n, 5, generator=g)
    labels[b, :n] = torch.stack([(r[:, 0] * 3).floor(), 20 + r[:, 1] * 600, 20 + r[:, 2] * 344, 16 + r[:, 3] * 200, 16 + r[:, 4] * 150], 1)
labels = labels.to(dev)


def step():
    feats, _ = model.forward_sequence(xs, None)
    sel = {}
    for s in (2, 3, 4):
        f = feats[s]                                                              # (T, B, C, H, W)-shaped view of channels-last storage
        fr = f.permute(0, 1, 3, 4, 2).reshape(T * B, f.shape[3], f.shape[4], f.shape[2])
        sel[s] = ops.gather_frames(fr, idx).permute(0, 3, 1, 2)
    det, losses = head(neck(sel), labels)
    losses['loss'].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return losses['loss']


for _ in range(3):
    loss = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
assert torch.isfinite(loss)
print(json.dumps({'what': 'detector training step: backbone (T=21, B=24, 1 Mpx) + PAFPN + head + SimOTA losses on K labelled frames + AdamW',
                  'K': K, 'dtype': 'bf16', 'ms_per_step': round(ms, 2), 'event_tensors_per_s': round(T * B / ms * 1e3, 1),
                  'backbone_only_ms_per_step_ref': 'bench.py (random cotangents on all frames)', 'loss': round(float(loss), 4)}), flush=True)
