import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from rvt_amd import RNNDetector, backbone_config, fpn as F_, head as H_, ops
from tests.test_detector_step import make_labels
dev = torch.device('cuda', 0)
T, B, K = 5, 4, 9
g = torch.Generator(device=dev).manual_seed(5)
xs = torch.randint(0, 11, (T, B, 20, 240, 304), generator=g, dtype=torch.uint8, device=dev)
sel = torch.linspace(0, T * B - 1, K).round().to(torch.int64).to(dev)
labels = make_labels(K, 8, (256, 320), 2, seed=3).to(dev)
out = {}
for dt in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    bb = RNNDetector(backbone_config('tiny', 'gen1'), compute_dtype=dt).to(dev)
    with torch.no_grad():
        for k, p in bb.named_parameters():
            if k.endswith('gamma'): p.fill_(0.5)
    dims, strides = bb.get_stage_dims((2, 3, 4)), bb.get_strides((2, 3, 4))
    torch.manual_seed(1)
    neck = F_.YOLOPAFPN(depth=0.33, in_channels=dims, compute_dtype=dt).to(dev).train()
    head = H_.YOLOXHead(num_classes=2, strides=strides, in_channels=dims, compute_dtype=dt).to(dev).train()
    feats, _ = bb.forward_sequence(xs, None)
    fsel = {}
    keep = {}
    for s in (2, 3, 4):
        f = feats[s]
        fr = f.permute(0, 1, 3, 4, 2).reshape(T * B, f.shape[3], f.shape[4], f.shape[2])
        fs = ops.gather_frames(fr, sel.to(torch.int32)).permute(0, 3, 1, 2)
        fs.retain_grad(); keep[s] = fs
        fsel[s] = fs
    _, losses = head(neck(fsel), labels)
    losses['loss'].backward()
    torch.cuda.synchronize()
    out[dt] = dict(loss=float(losses['loss'].detach()), nfg=float(losses['num_fg']), feat={s: keep[s].detach().double() for s in keep},
                   dfeat={s: keep[s].grad.double() for s in keep}, g={k: p.grad.double().flatten() for k, p in bb.named_parameters()},
                   match=head.last_match.clone())
a, b = out[torch.float32], out[torch.bfloat16]
print('loss', a['loss'], b['loss'], 'num_fg', a['nfg'], b['nfg'], 'anchors matched differently', int((a['match'] != b['match']).sum()), 'of', a['match'].numel())
cos = lambda x, y: float((x * y).sum() / (x.norm() * y.norm()))
for s in (2, 3, 4):
    print('stage', s, 'feat rel', float((a['feat'][s] - b['feat'][s]).norm() / a['feat'][s].norm()), 'dfeat cos', cos(a['dfeat'][s].flatten(), b['dfeat'][s].flatten()),
          'norm ratio', float(b['dfeat'][s].norm() / a['dfeat'][s].norm()))
for st in range(4):
    ka = [k for k in a['g'] if k.startswith(f'stages.{st}.')]
    x, y = torch.cat([a['g'][k] for k in ka]), torch.cat([b['g'][k] for k in ka])
    print('stage', st + 1, 'param grad cos', cos(x, y), 'norm', float(x.norm()), float(y.norm()))
worst = sorted(((cos(a['g'][k], b['g'][k]), k, float(a['g'][k].norm())) for k in a['g']))[:12]
for w in worst: print(w)
