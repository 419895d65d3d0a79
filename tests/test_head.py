"""YOLOX head + SimOTA + detection losses: rvt_amd.head on the HIP kernels (emulator build on CPU,
gfx950 build on the GPU) and the CPU oracle, both against fixtures recorded from the unmodified reference
(oracle/make_golden_head.py).

Bars.  The assignment (matched ground truth of every anchor) is integer work: bit-exact against the reference.  fp32 floating point:
1e-3 of the tensor scale (the north_star tolerance; the tail alone measures ~1e-6).  bf16: the towers run in bf16, so the predictions
move by ~1e-2 and an anchor near a cost tie may legitimately change hands; the bf16 test therefore checks the detections against the
reference within 8e-2 and the TAIL against the oracle run on the head's own bf16 prediction maps (exact assignment, 1e-3 losses)."""
import numpy as np
import pytest
import torch

from rvt_amd import head as H_
from tests import casegen_head as cg
from tests.backends import backend  # noqa: F401
from tests.harness import load_golden
from tests.test_fpn import _check_grads, _rel


def _maps(name, dev, grad=False):
    maps_np, lab = cg.make_pred_maps(name)
    maps = [torch.from_numpy(m).to(dev).requires_grad_(grad) for m in maps_np]
    return maps, torch.from_numpy(lab).to(dev)


@pytest.mark.parametrize('name', list(cg.SIMOTA_CASES))
def test_head_oracle_matches_reference_golden(name):
    """Pins oracle/head_oracle.py (decode, SimOTA, losses, gradient) to the reference (CPU only)."""
    from oracle import head_oracle as O
    c, gold = cg.SIMOTA_CASES[name], load_golden(name)
    maps, labels = _maps(name, 'cpu', grad=True)
    hws = cg.level_hws(c)
    pred = O.decode_train(maps, hws, c['strides'], c['nc'])
    losses, match, piou = O.head_losses(pred, labels, hws, c['strides'], c['nc'])
    losses[0].backward()
    assert np.array_equal(match.numpy(), gold['match'])
    assert _rel(piou.numpy(), gold['piou']) <= 1e-5
    assert _rel(O.to_infer(pred).detach().numpy(), gold['detections']) <= 1e-6
    assert np.abs(losses.detach().numpy() - gold['losses']).max() <= 1e-5 * np.abs(gold['losses']).max()
    for i, m in enumerate(maps):
        assert _rel(m.grad.numpy(), gold[f'dmap{i}']) <= 1e-4, f'map {i}'


@pytest.mark.parametrize('name', list(cg.SIMOTA_CASES))
def test_simota_tail_vs_reference_golden(backend, name):
    """decode + batched on-device SimOTA + losses + gradient on crafted prediction maps: dense scenes (dynamic k up to 10, anchors
    claimed by several ground truths), an image without labels, boxes at the border, the full 1 Mpx anchor count (A = 5040)."""
    dev = backend
    c, gold = cg.SIMOTA_CASES[name], load_golden(name)
    maps, labels = _maps(name, dev, grad=True)
    det, losses, match, piou = H_.simota_loss(maps, labels, cg.level_hws(c), c['strides'], c['nc'])
    losses[0].backward()
    assert np.array_equal(match.cpu().numpy(), gold['match']), \
        f'{int((match.cpu().numpy() != gold["match"]).sum())} anchors assigned differently from the reference'
    assert _rel(piou.cpu().numpy(), gold['piou']) <= 1e-5
    assert _rel(det.cpu().numpy(), gold['detections']) <= 1e-5
    err = np.abs(losses.detach().cpu().numpy() - gold['losses']).max() / np.abs(gold['losses']).max()
    assert err <= 1e-5, f'losses {losses.tolist()} vs {gold["losses"].tolist()}'
    for i, m in enumerate(maps):
        e = _rel(m.grad.cpu().numpy(), gold[f'dmap{i}'])
        assert e <= 1e-4, f'gradient of map {i}: rel err {e:.3e}'
        if i % 2 == 0:
            assert float(m.grad[..., 5:].abs().max()) == 0.0            # padding columns of the 8-aligned prediction GEMM
    # every loss component is differentiable on its own: d(5 iou)/d box columns only, d obj / d cls likewise
    maps2, _ = _maps(name, dev, grad=True)
    _, l2, _, _ = H_.simota_loss(maps2, labels, cg.level_hws(c), c['strides'], c['nc'])
    (l2[1] + l2[2] + l2[3]).backward()
    for a, b in zip(maps, maps2):
        assert _rel(b.grad.cpu().numpy(), a.grad.cpu().numpy()) <= 1e-6


def test_simota_no_labels_and_properties(backend):
    """Empty label tensors (G = 0 and all-zero rows) give the pure background loss; permuting the images permutes the assignment;
    a second identical call is bit-identical (no float atomics anywhere in the tail)."""
    dev = backend
    name = 'simota_small'
    c = cg.SIMOTA_CASES[name]
    hws, st, nc = cg.level_hws(c), c['strides'], c['nc']
    maps, labels = _maps(name, dev)
    det, l0, m0, _ = H_.simota_loss(maps, labels, hws, st, nc)
    det1, l1, m1, _ = H_.simota_loss(maps, labels, hws, st, nc)
    assert torch.equal(l0, l1) and torch.equal(m0, m1) and torch.equal(det, det1)
    for lab in (labels[:, :0], torch.zeros_like(labels)):
        _, l, m, _ = H_.simota_loss(maps, lab, hws, st, nc)
        assert int((m >= 0).sum()) == 0
        pred_obj = torch.cat([mm.reshape(mm.shape[0], -1, mm.shape[-1])[..., 4] for mm in maps[0::2]], 1)
        want = torch.nn.functional.binary_cross_entropy_with_logits(pred_obj, torch.zeros_like(pred_obj), reduction='sum')
        assert abs(float(l[2]) - float(want)) <= 1e-5 * float(want) and float(l[1]) == 0.0 and float(l[3]) == 0.0
        assert float(l[4]) == 1.0                                        # max(num_fg, 1) / max(num_gts, 1), yolo_head.py:413,442
    perm = [2, 0, 1]
    _, lp, mp, _ = H_.simota_loss([m[perm] for m in maps], labels[perm], hws, st, nc)
    assert torch.equal(mp, m0[perm])
    assert abs(float(lp[0]) - float(l0[0])) <= 1e-5 * float(l0[0])


def _shapes(name, gold):
    c = cg.CASES[name]
    m = H_.YOLOXHead(num_classes=c['nc'], strides=c['strides'], in_channels=c['in_channels'])
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert [k for k, _ in shapes] == [str(n) for n in gold['names']], 'parameter / buffer names differ from the reference state_dict'
    return shapes


def _build(name, dev, dtype):
    c, gold = cg.CASES[name], load_golden(name)
    m = H_.YOLOXHead(num_classes=c['nc'], strides=c['strides'], in_channels=c['in_channels'], compute_dtype=dtype)
    sd = {k: torch.from_numpy(v) for k, v in cg.make_params(name, _shapes(name, gold)).items()}
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return m.to(dev), gold


@pytest.mark.parametrize('name', list(cg.CASES))
def test_head_fp32_vs_reference_golden(backend, name):
    dev = backend
    if name == 'head_base' and dev.type == 'cpu':
        pytest.skip('RVT-Base widths: GPU only (minutes on the CPU emulator)')
    c = cg.CASES[name]
    m, gold = _build(name, dev, torch.float32)
    xs = [torch.from_numpy(a).to(dev) for a in cg.make_inputs(name)]
    labels = torch.from_numpy(cg.make_labels(name, c)).to(dev)
    m.eval()
    with torch.no_grad():
        det, none = m(xs)
    assert none is None and _rel(det.cpu().numpy(), gold['eval_detections']) <= 1e-3
    m.train()
    xg = [x.clone().requires_grad_(True) for x in xs]
    det, losses = m(xg, labels)
    assert set(losses) == {'loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg'} and losses['l1_loss'] == 0.0
    losses['loss'].backward()
    assert _rel(det.cpu().numpy(), gold['train_detections']) <= 1e-3
    assert np.array_equal(m.last_match.cpu().numpy(), gold['match'])          # the same anchors, the same ground truths
    assert _rel(m.last_matched_iou.cpu().numpy(), gold['piou']) <= 1e-3
    got = np.array([float(losses[k].detach()) for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'num_fg')])
    assert np.abs(got - gold['losses']).max() <= 1e-3 * np.abs(gold['losses']).max(), f'{got} vs {gold["losses"]}'
    for i, x in enumerate(xg):
        e = _rel(x.grad.cpu().numpy(), gold[f'dx{i}'])
        assert e <= 1e-3, f'input gradient of level {i}: rel err {e:.3e}'
    grads = dict(m.named_parameters())
    worst = _check_grads(lambda k: grads[k].grad.float().cpu().numpy(), gold, 1e-3, f'hip [{name}]')
    for k, b in m.named_buffers():
        if not k.endswith('num_batches_tracked'):
            assert _rel(b.float().cpu().numpy(), gold['buf/' + k]) <= 1e-3, k
    print(f'{name}: worst gradient err {worst:.3e}')


@pytest.mark.parametrize('name', list(cg.CASES))
def test_head_bf16(backend, name):
    """bf16 towers: detections against the reference within bf16 accuracy; the tail against the oracle on the head's OWN maps."""
    from oracle import head_oracle as O
    dev = backend
    if name == 'head_base' and dev.type == 'cpu':
        pytest.skip('RVT-Base widths: GPU only (minutes on the CPU emulator)')
    c = cg.CASES[name]
    m, gold = _build(name, dev, torch.bfloat16)
    xs = [torch.from_numpy(a).to(dev) for a in cg.make_inputs(name)]
    labels = torch.from_numpy(cg.make_labels(name, c)).to(dev)
    m.eval()
    with torch.no_grad():
        det, _ = m(xs)
    assert _rel(det.cpu().numpy(), gold['eval_detections']) <= 8e-2
    m.train()
    maps, hws = m._pred_maps(xs)
    maps = [t.detach() for t in maps]
    det, ls, match, piou = H_.simota_loss([t.clone().requires_grad_(True) for t in maps], labels, hws, c['strides'], c['nc'])
    ref_pred = O.decode_train([t.float().cpu() for t in maps], hws, c['strides'], c['nc'])
    want, wmatch, wpiou = O.head_losses(ref_pred, labels.cpu(), hws, c['strides'], c['nc'])
    assert np.array_equal(match.cpu().numpy(), wmatch.numpy())
    assert _rel(piou.cpu().numpy(), wpiou.numpy()) <= 1e-5
    assert np.abs(ls.detach().cpu().numpy() - want.numpy()).max() <= 1e-4 * float(want.abs().max())
    # and the whole training step runs and lands near the fp32 reference losses
    xg = [x.clone().requires_grad_(True) for x in xs]
    det, losses = m(xg, labels)
    losses['loss'].backward()
    assert all(torch.isfinite(x.grad).all() for x in xg)
    assert abs(float(losses['loss'].detach()) - gold['losses'][0]) <= 0.15 * gold['losses'][0]


def test_head_rejects_what_is_not_built():
    with pytest.raises(NotImplementedError):
        H_.YOLOXHead(num_classes=3, in_channels=(32, 64, 128), depthwise=True)
    m = H_.YOLOXHead(num_classes=3, in_channels=(32, 64, 128))
    m.decode_in_inference = False
    with pytest.raises(NotImplementedError):
        m.eval()([torch.zeros(1, 32, 4, 4)])


@pytest.mark.parametrize('seed,nc,hws,strides,N,G', [
    (0, 1, ((5, 7),), (8,), 2, 3),                                          # one level, one class
    (1, 80, ((8, 8), (4, 4), (2, 2)), (8, 16, 32), 2, 9),                   # COCO-sized class vector (padded prediction GEMM columns)
    (2, 3, ((16, 12), (8, 6), (4, 3), (2, 2)), (4, 8, 16, 32), 3, 40),      # four levels, more boxes than any image fills
    (3, 2, ((3, 5), (2, 3)), (16, 32), 5, 1),                               # at most one box per image
])
def test_simota_tail_random_shapes_vs_oracle(backend, seed, nc, hws, strides, N, G):
    """Shapes the reference fixtures do not cover, against oracle/head_oracle.py (pinned to the reference above): assignment exact,
    losses and the gradient into the prediction maps within fp32 round-off."""
    from oracle import head_oracle as O
    dev = backend
    g = torch.Generator().manual_seed(100 + seed)
    Hi, Wi = hws[0][0] * strides[0], hws[0][1] * strides[0]
    labels = torch.zeros(N, G, 5)
    for b in range(N):
        n = int(torch.randint(0, G + 1, (1,), generator=g))
        r = torch.rand(n, 5, generator=g)
        labels[b, :n] = torch.stack([(r[:, 0] * nc).floor(), 1 + r[:, 1] * (Wi - 2), 1 + r[:, 2] * (Hi - 2),
                                     4 + r[:, 3] * 0.5 * Wi, 4 + r[:, 4] * 0.5 * Hi], 1)
    ncp = (nc + 7) // 8 * 8
    maps = []
    for (h, w) in hws:
        ro = torch.zeros(N, h, w, 8)
        ro[..., :5] = torch.randn(N, h, w, 5, generator=g) * torch.tensor([0.5, 0.5, 0.7, 0.7, 2.0])
        ro[..., 2:4] += 0.8
        cl = torch.zeros(N, h, w, ncp)
        cl[..., :nc] = torch.randn(N, h, w, nc, generator=g) * 2.0
        maps += [ro, cl]
    mh = [m.clone().to(dev).requires_grad_(True) for m in maps]
    det, ls, match, piou = H_.simota_loss(mh, labels.to(dev), hws, strides, nc)
    ls[0].backward()
    mo = [m.clone().requires_grad_(True) for m in maps]
    pred = O.decode_train(mo, hws, strides, nc)
    want, wmatch, wpiou = O.head_losses(pred, labels, hws, strides, nc)
    want[0].backward()
    assert np.array_equal(match.cpu().numpy(), wmatch.numpy())
    assert _rel(piou.cpu().numpy(), wpiou.numpy()) <= 1e-5 or float(wpiou.abs().max()) == 0.0
    assert float((ls.detach().cpu() - want.detach()).abs().max()) <= 1e-5 * float(want.detach().abs().max())
    assert _rel(det.cpu().numpy(), O.to_infer(pred).detach().numpy()) <= 1e-5
    for a, b in zip(mh, mo):
        if float(b.grad.abs().max()) > 0:
            assert _rel(a.grad.cpu().numpy(), b.grad.numpy()) <= 2e-4
        else:
            assert float(a.grad.abs().max()) == 0.0


def test_head_eval_caches_follow_parameter_updates(backend):
    """Inference caches (padded prediction weights, the BatchNorm affine of every unit) are keyed on the parameters' versions."""
    dev = backend
    m, _ = _build('head_micro', dev, torch.float32)
    m.eval()
    xs = [torch.from_numpy(a).to(dev) for a in cg.make_inputs('head_micro')]
    with torch.no_grad():
        a, _ = m(xs)
        b, _ = m(xs)
        assert torch.equal(a, b)
        m.cls_preds[0].bias.add_(1.0)
        c, _ = m(xs)
        assert float((c[:, :240, 5:] - a[:, :240, 5:]).abs().min()) > 0 and torch.equal(c[:, 240:], a[:, 240:])   # level 0 = the first 12 x 20 anchors
        m.stems[2].bn.weight.mul_(0.5)
        d, _ = m(xs)
        assert not torch.equal(d[:, 300:], c[:, 300:]) and torch.equal(d[:, :300], c[:, :300])                     # level 2 = the last 3 x 5 anchors
