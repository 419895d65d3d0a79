"""The WHOLE detector training step (backbone over the sequence -> gather of the labelled frames -> YOLOX PAFPN -> YOLOX head ->
SimOTA losses -> backward through everything), reference call stack modules/detection.py:104-206 with train.py:60-67,133 (DDP +
SyncBatchNorm):

  * world 2 over gloo (CPU, kernel sources on the SIMT emulator): every rank runs its own sequences; the backbone's gradients go
    through StageGradReducer (per-stage buckets, averaged), the PAFPN / head gradients through a plain all-reduce, BatchNorm statistics
    are synchronised - loss, EVERY parameter gradient and the BatchNorm running statistics must equal ONE process on the concatenated
    batch whose loss is the mean of the per-rank losses (each rank normalises its SimOTA losses by its own foreground count, exactly
    as the reference does under DDP);
  * GPU: the step on the production route in bf16 against the fp32 route of the same code (the anchor for
    profiles/bench_detector_step.py's timing).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_detector(name, device, dtype, nc=3, seed=0):
    from rvt_amd import fpn as F_, head as H_
    from tests.test_backbone import build_model
    bb = build_model(name, device, dtype)
    dims, strides = bb.get_stage_dims((2, 3, 4)), bb.get_strides((2, 3, 4))
    torch.manual_seed(seed)
    neck = F_.YOLOPAFPN(depth=0.33, in_channels=dims, compute_dtype=dtype)
    head = H_.YOLOXHead(num_classes=nc, strides=strides, in_channels=dims, compute_dtype=dtype)
    with torch.no_grad():                               # BatchNorm affine away from (1, 0) so that its gradients are exercised
        g = torch.Generator().manual_seed(seed + 1)
        for m in list(neck.modules()) + list(head.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    return bb, neck.to(device).train(), head.to(device).train()


def make_labels(K, G, hw, nc, seed):
    r = np.random.default_rng(seed)
    H, W = hw
    lab = np.zeros((K, G, 5), dtype=np.float32)
    for b in range(K):
        n = int(r.integers(0 if b % 3 == 2 else 1, G + 1))
        for j in range(n):
            lab[b, j] = [float(r.integers(0, nc)), r.uniform(0.1 * W, 0.9 * W), r.uniform(0.1 * H, 0.9 * H),
                         r.uniform(0.1, 0.5) * W, r.uniform(0.1, 0.5) * H]
    return torch.from_numpy(lab)


def detector_loss(bb, neck, head, xs, sel, labels, groups=None):
    """xs (T, B, 20, h, w) uint8 / float; sel: indices t * B + b of the labelled frames; labels [K][G][5].  groups: list of index
    lists into the K frames - the loss is the MEAN over the groups of the SimOTA loss of each group (one group = plain step)."""
    from rvt_amd import head as H_, ops
    feats, _ = bb.forward_sequence(xs, None)
    T, B = xs.shape[:2]
    fsel = {}
    for s in (2, 3, 4):
        f = feats[s]                                    # (T, B, C, H, W)-shaped view of channels-last storage
        fr = f.permute(0, 1, 3, 4, 2).reshape(T * B, f.shape[3], f.shape[4], f.shape[2])
        fsel[s] = ops.gather_frames(fr, sel.to(torch.int32)).permute(0, 3, 1, 2)
    if groups is None:
        _, losses = head(neck(fsel), labels)
        return losses['loss']
    maps, hws = head._pred_maps(neck(fsel))
    total = 0.0
    for gidx in groups:
        gi = torch.as_tensor(gidx, dtype=torch.long, device=labels.device)
        _, ls, _, _ = H_.simota_loss([m.index_select(0, gi) for m in maps], labels.index_select(0, gi), hws, head.strides, head.num_classes)
        total = total + ls[0]
    return total / len(groups)


DDP_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
root, ref_path = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from rvt_amd import _lib, tuning
tuning.use(**dict(tuning.TEST_GEOMETRY, gemm_resident=3))
from rvt_amd.dist import StageGradReducer
from tests.backends import emu_library
from tests import casegen
from tests.test_detector_step import build_detector, detector_loss, make_labels
_lib._install_test_library(emu_library())
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
torch.set_num_threads(2)
ref = torch.load(ref_path)
bb, neck, head = build_detector('micro', torch.device('cpu'), torch.float32)
red = StageGradReducer().attach(bb)
xs = torch.from_numpy(casegen.make_inputs('micro'))[:, rank:rank + 1].contiguous()        # this rank's sequence
T = xs.shape[0]
sel_full, labels_full = ref['sel'], ref['labels']
mine = [i for i in range(len(sel_full)) if int(sel_full[i]) % world == rank]             # labelled frames of this rank's sample
sel = torch.tensor([int(sel_full[i]) // world for i in mine])                             # (t, b) -> t * 1 + 0 in the local batch
loss = detector_loss(bb, neck, head, xs, sel, labels_full[mine])
loss.backward()
red.finish()
tail = [p for m in (neck, head) for p in m.parameters()]
for p in tail:                                                                            # "FPN / head parameters: a plain all-reduce"
    dist.all_reduce(p.grad)
    p.grad /= world
lsum = loss.detach().clone()
dist.all_reduce(lsum)
assert abs(float(lsum) / world - float(ref['loss'])) <= 1e-4 * abs(float(ref['loss'])), (float(lsum) / world, float(ref['loss']))
worst = 0.0
for tag, m in (('bb', bb), ('neck', neck), ('head', head)):
    for k, p in m.named_parameters():
        want = ref['grad'][tag + '/' + k]
        err = float((p.grad - want).abs().max()) / max(float(want.abs().max()), 1e-12)
        worst = max(worst, err)
        assert err < 2e-3, (tag, k, err)
    for k, b in m.named_buffers():
        if b.dtype.is_floating_point:
            want = ref['buf'][tag + '/' + k]
            err = float((b - want).abs().max()) / max(float(want.abs().max()), 1e-12)
            assert err < 1e-4, (tag, k, err)
dist.destroy_process_group()
print('OK', rank, 'worst gradient err %.2e' % worst)
'''


def test_detector_step_data_parallel_world2_gloo(tmp_path):
    from rvt_amd import _lib, tuning
    from tests import casegen
    from tests.backends import emu_library
    lib = emu_library()
    saved = tuning.overrides()
    _lib._install_test_library(lib)
    try:
        tuning.use(**dict(tuning.TEST_GEOMETRY, gemm_resident=3))
        bb, neck, head = build_detector('micro', torch.device('cpu'), torch.float32)
        xs = torch.from_numpy(casegen.make_inputs('micro'))                               # (T, B = 2, 20, h, w)
        T, B = xs.shape[:2]
        sel = torch.tensor([t * B + b for t in (0, T - 1) for b in range(B)] + [1 * B + 0])   # labelled frames, uneven over the samples
        labels = make_labels(len(sel), 5, tuple(bb.in_res_hw), 3, seed=11)
        groups = [[i for i in range(len(sel)) if int(sel[i]) % B == r] for r in range(B)]   # rank r = sample r
        loss = detector_loss(bb, neck, head, xs, sel, labels, groups=groups)
        loss.backward()
        ref = dict(loss=loss.detach(), sel=sel, labels=labels,
                   grad={tag + '/' + k: p.grad.clone() for tag, m in (('bb', bb), ('neck', neck), ('head', head)) for k, p in m.named_parameters()},
                   buf={tag + '/' + k: b.clone() for tag, m in (('bb', bb), ('neck', neck), ('head', head)) for k, b in m.named_buffers()})
        assert all(torch.isfinite(g).all() for g in ref['grad'].values()) and float(loss.detach()) > 0
    finally:
        tuning.use(**saved)
        _lib._install_test_library(None)
    ref_path = tmp_path / 'ref.pt'
    torch.save(ref, ref_path)
    script = tmp_path / 'ddp.py'
    script.write_text(DDP_WORKER)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', '29747', str(script), ROOT, str(ref_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count('OK') == 2, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_detector_step_production_route_bf16_vs_fp32(production_route):
    """RVT-Tiny / Gen1 detector step, B = 4, T = 5, K = 9 labelled frames, on the library defaults: the bf16 step against the fp32 step of
    the same code.  Measured on MI355X (profiles/diag_detector_bf16.py): backbone features agree to 0.5 - 0.9 % and the loss to 0.2 %, 2 of
    15 120 anchors are matched differently, but the cotangent the TAIL hands back to the features only agrees to a cosine of 0.82 - 0.87
    (twenty conv + batch-statistics BatchNorm layers in 8 mantissa bits, statistics over nine frames), and the backbone's parameter
    gradients inherit exactly that (0.83 - 0.87 per stage).  The bars are therefore: tight on loss and assignment, and a direction /
    norm check on the gradients that still catches a wrong sign, a missing term or a batch-size factor."""
    from rvt_amd import RNNDetector, backbone_config, fpn as F_, head as H_
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
        with torch.no_grad():                           # LayerScale O(1): at the 1e-5 default the attention / MLP branches are invisible
            for k, p in bb.named_parameters():
                if k.endswith('gamma'):
                    p.fill_(0.5)
        dims, strides = bb.get_stage_dims((2, 3, 4)), bb.get_strides((2, 3, 4))
        torch.manual_seed(1)
        neck = F_.YOLOPAFPN(depth=0.33, in_channels=dims, compute_dtype=dt).to(dev).train()
        head = H_.YOLOXHead(num_classes=2, strides=strides, in_channels=dims, compute_dtype=dt).to(dev).train()
        loss = detector_loss(bb, neck, head, xs, sel, labels)
        loss.backward()
        torch.cuda.synchronize()
        out[dt] = (float(loss.detach()), {k: p.grad.double().flatten() for k, p in bb.named_parameters()},
                   {k: p.grad.double().flatten() for m in (neck, head) for k, p in m.named_parameters()}, head.last_match.clone())
    l32, g32, t32, m32 = out[torch.float32]
    l16, g16, t16, m16 = out[torch.bfloat16]
    assert np.isfinite(l32) and np.isfinite(l16) and l32 > 0
    assert abs(l16 - l32) <= 3e-2 * abs(l32), (l16, l32)
    assert int((m32 != m16).sum()) <= 0.01 * m32.numel(), int((m32 != m16).sum())
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
    report = []
    for st in range(4):
        ks = [k for k in g32 if k.startswith(f'stages.{st}.')]
        a, b = torch.cat([g32[k] for k in ks]), torch.cat([g16[k] for k in ks])
        c, r = cos(a, b), float(b.norm() / a.norm())
        report.append(f'stage {st + 1}: cos {c:.3f} norm ratio {r:.3f}')
        assert c >= 0.70 and 0.75 <= r <= 1.25, report[-1]
    a, b = torch.cat(list(t32.values())), torch.cat(list(t16.values()))
    c_t = cos(a, b)
    assert c_t >= 0.70 and 0.75 <= float(b.norm() / a.norm()) <= 1.25, (c_t, float(b.norm() / a.norm()))
    print(f'detector step bf16 vs fp32: loss {l16:.5f} / {l32:.5f}; backbone gradients ' + '; '.join(report) + f'; tail gradient cosine {c_t:.3f}')
