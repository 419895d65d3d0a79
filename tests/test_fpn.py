"""YOLOX PAFPN: rvt_amd.fpn on the HIP kernels (emulator build on CPU, gfx950 build on the GPU)
and the CPU oracle, both against fixtures recorded from the unmodified reference (oracle/make_golden_fpn.py): eval-mode forward
(running statistics), training-mode forward + backward (every parameter gradient, the input gradients, the running-statistics
update).  fp32 bar 1e-3 of the tensor scale (the north_star tolerance; measured 1.4e-5 on the gradients at RVT-Base widths); bf16 against the
fp32 reference: 8e-2 on outputs / running statistics, 1.5e-1 on the worst element of a gradient relative to its max (twenty
conv + batch-statistics BatchNorm layers deep in 8 mantissa bits, statistics over as few as 30 rows in these cases; measured worst
5.0e-2 / 6.0e-2 on MI355X)."""
import numpy as np
import pytest
import torch

from rvt_amd import fpn as F_
from tests import casegen_fpn as cg
from tests.backends import backend  # noqa: F401
from tests.harness import load_golden


def _rel(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def _check_grads(get_grad, gold, tol, what):
    worst = 0.0
    for k in gold:
        if k.startswith('grad/'):
            err = _rel(get_grad(k[5:]), gold[k])
        elif k.startswith('gradstat/'):
            g = np.asarray(get_grad(k[9:]), dtype=np.float64).reshape(-1)
            err = abs(np.sqrt((g * g).sum()) - gold[k][0]) / max(gold[k][0], 1e-30)
            samp = g[np.linspace(0, g.size - 1, 512).astype(np.int64)]
            err = max(err, _rel(samp, gold['gradsamp/' + k[9:]]))
        else:
            continue
        worst = max(worst, err)
        assert err <= tol, f'{what}: gradient of {k.split("/", 1)[1]}: rel err {err:.3e} > {tol:.1e}'
    return worst


@pytest.mark.parametrize('name', list(cg.CASES))
def test_fpn_oracle_matches_reference_golden(name):
    """Pins oracle/fpn_oracle.py to the reference (CPU only)."""
    from oracle import fpn_oracle as O
    c, gold = cg.CASES[name], load_golden(name)
    p = {k: torch.from_numpy(v) for k, v in cg.make_params(name, _shapes(name, gold)).items()}
    xs = {s: torch.from_numpy(a) for s, a in cg.make_inputs(name).items()}
    cots = [torch.from_numpy(a) for a in cg.make_cotangents(name)]
    outs, _ = O.pafpn_forward(xs, p, c['depth'], training=False)
    for i, o in enumerate(outs):
        assert _rel(o.numpy(), gold[f'eval_out{i}']) <= 2e-4
    pg = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and 'running' not in k else v) for k, v in p.items()}
    xg = {s: x.clone().requires_grad_(True) for s, x in xs.items()}
    outs, ns = O.pafpn_forward(xg, pg, c['depth'], training=True)
    loss = sum((o * ct).sum() for o, ct in zip(outs, cots))
    loss.backward()
    for i, o in enumerate(outs):
        assert _rel(o.detach().numpy(), gold[f'train_out{i}']) <= 2e-4
    for s in (2, 3, 4):
        assert _rel(xg[s].grad.numpy(), gold[f'dx{s}']) <= 5e-4
    _check_grads(lambda k: pg[k].grad.numpy(), gold, 5e-4, f'oracle [{name}]')
    for k, v in ns.items():
        assert _rel(v.numpy(), gold['buf/' + k]) <= 2e-4, k


def _shapes(name, gold):
    """state_dict names + shapes of the case, from the module itself (asserted equal to the reference's by make_golden_fpn.py)."""
    c = cg.CASES[name]
    m = F_.YOLOPAFPN(depth=c['depth'], in_channels=c['in_channels'])
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert [k for k, _ in shapes] == [str(n) for n in gold['names']], 'parameter / buffer names differ from the reference state_dict'
    return shapes


def _build(name, dev, dtype):
    c, gold = cg.CASES[name], load_golden(name)
    m = F_.YOLOPAFPN(depth=c['depth'], in_channels=c['in_channels'], compute_dtype=dtype)
    sd = {k: torch.from_numpy(v) for k, v in cg.make_params(name, _shapes(name, gold)).items()}
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return m.to(dev), gold


@pytest.mark.parametrize('name,dtype,tol,gtol', [('fpn_micro', torch.float32, 1e-3, 1e-3), ('fpn_base', torch.float32, 1e-3, 1e-3),
                                                 ('fpn_micro', torch.bfloat16, 8e-2, 1.5e-1), ('fpn_base', torch.bfloat16, 8e-2, 1.5e-1)])
def test_fpn_hip_vs_reference_golden(backend, name, dtype, tol, gtol):
    dev = backend
    if name == 'fpn_base' and dev.type == 'cpu':
        pytest.skip('RVT-Base widths: GPU only (minutes on the CPU emulator)')
    m, gold = _build(name, dev, dtype)
    xs = {s: torch.from_numpy(a).to(dev) for s, a in cg.make_inputs(name).items()}
    cots = [torch.from_numpy(a).to(dev) for a in cg.make_cotangents(name)]
    m.eval()
    with torch.no_grad():
        outs = m(xs)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(gold[f'eval_out{i}'].shape)
        err = _rel(o.float().cpu().numpy(), gold[f'eval_out{i}'])
        assert err <= tol, f'eval forward, output {i}: rel err {err:.3e}'
    m.train()
    xg = {s: x.clone().requires_grad_(True) for s, x in xs.items()}
    outs = m(xg)
    loss = sum((o.float() * ct).sum() for o, ct in zip(outs, cots))
    loss.backward()
    for i, o in enumerate(outs):
        err = _rel(o.detach().float().cpu().numpy(), gold[f'train_out{i}'])
        assert err <= tol, f'training forward, output {i}: rel err {err:.3e}'
    for s in (2, 3, 4):
        err = _rel(xg[s].grad.float().cpu().numpy(), gold[f'dx{s}'])
        assert err <= gtol, f'input gradient of stage {s}: rel err {err:.3e}'
    grads = dict(m.named_parameters())
    worst = _check_grads(lambda k: grads[k].grad.float().cpu().numpy(), gold, gtol, f'hip [{name}, {dtype}]')
    for k, b in m.named_buffers():
        if k.endswith('num_batches_tracked'):
            assert int(b) == 1
        else:
            err = _rel(b.float().cpu().numpy(), gold['buf/' + k])
            assert err <= tol, f'running statistic {k}: rel err {err:.3e}'
    print(f'{name} {dtype}: worst gradient err {worst:.3e}')


def test_fpn_rejects_what_is_not_built():
    with pytest.raises(NotImplementedError):
        F_.YOLOPAFPN(depthwise=True)
    with pytest.raises(NotImplementedError):
        F_.BaseConv(8, 8, 3, 1, act='relu')


SYNCBN_WORKER = r'''
# Data-parallel PAFPN: each rank holds half of the batch; the BatchNorm statistics are all-reduced (SyncBatchNorm, reference
# train.py:133), so outputs / input gradients of the local samples and the SUM over ranks of the parameter gradients must equal the
# single-process full-batch fixture recorded from the reference.
import os, sys, numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root)
from rvt_amd import _lib, tuning
from tests.backends import emu_library
from tests import casegen_fpn as cg
from tests.harness import load_golden
from tests.test_fpn import _build, _rel, _check_grads
tuning.use(**tuning.TEST_GEOMETRY)
_lib._install_test_library(emu_library())
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
torch.set_num_threads(2)
name = 'fpn_micro'
m, gold = _build(name, torch.device('cpu'), torch.float32)
xs = {s: torch.from_numpy(a)[rank:rank + 1].clone().requires_grad_(True) for s, a in cg.make_inputs(name).items()}
cots = [torch.from_numpy(a)[rank:rank + 1] for a in cg.make_cotangents(name)]
m.train()
outs = m(xs)
sum((o * ct).sum() for o, ct in zip(outs, cots)).backward()
for i, o in enumerate(outs):
    assert _rel(o.detach().numpy(), gold[f'train_out{i}'][rank:rank + 1]) <= 1e-3, ('out', i)
for s in (2, 3, 4):
    assert _rel(xs[s].grad.numpy(), gold[f'dx{s}'][rank:rank + 1]) <= 1e-3, ('dx', s)
grads = {}
for k, p in m.named_parameters():
    g = p.grad.clone()
    dist.all_reduce(g)
    grads[k] = g.numpy()
_check_grads(lambda k: grads[k], gold, 1e-3, 'syncbn')
for k, b in m.named_buffers():
    if not k.endswith('num_batches_tracked'):
        assert _rel(b.numpy(), gold['buf/' + k]) <= 1e-3, k
dist.destroy_process_group()
print('OK', rank)
'''


def test_fpn_syncbn_world2_gloo(tmp_path):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'w.py'
    script.write_text(SYNCBN_WORKER)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', '29743', str(script), root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count('OK') == 2, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.bfloat16, 3e-2)])
def test_fpn_eval_fused_conv_bn_silu_matches_unfused(backend, dtype, tol):
    """Inference takes ONE launch per BaseConv (BatchNorm affine + SiLU in the convolution's epilogue, rvt_conv_bn_act_fwd); an input that
    requires grad takes the three-launch route (conv, finalize, bn_act): same numbers up to the rounding of the intermediate."""
    dev = backend
    m, _ = _build('fpn_micro', dev, dtype)
    m.eval()
    xs = {s: torch.from_numpy(a).to(dev) for s, a in cg.make_inputs('fpn_micro').items()}
    with torch.no_grad():
        fused = m(xs)
    assert all(c._pk is not None and c._pk.fin is not None for c in m._pack.convs)
    unfused = m({s: x.clone().requires_grad_(True) for s, x in xs.items()})
    for a, b in zip(fused, unfused):
        assert _rel(a.float().cpu().numpy(), b.detach().float().cpu().numpy()) <= tol
    # a parameter change is picked up (the cached affine is keyed on the BatchNorm tensors' versions)
    with torch.no_grad():
        m.lateral_conv0.bn.running_var.mul_(4.0)
        changed = m(xs)
    assert _rel(changed[2].float().cpu().numpy(), fused[2].float().cpu().numpy()) > 1e-3


def test_fpn_gradient_accumulation_over_micro_batches(backend):
    """Two backward passes without zeroing in between must ADD (the weight gradient of a 1 x 1 BaseConv was a view of the
    ConvPack scratch arena that the next forward zeroes - the accumulated .grad was wiped and the second pass counted twice)."""
    dev = backend
    m, _ = _build('fpn_micro', dev, torch.float32)
    m.train()
    xs = {s: torch.from_numpy(a).to(dev) for s, a in cg.make_inputs('fpn_micro').items()}
    cots = [torch.from_numpy(a).to(dev) for a in cg.make_cotangents('fpn_micro')]

    def run(scale):
        outs = m({s: x * scale for s, x in xs.items()})
        sum((o.float() * ct).sum() for o, ct in zip(outs, cots)).backward()
    single = []
    for scale in (1.0, 0.5):
        m.zero_grad(set_to_none=True)
        run(scale)
        single.append({k: p.grad.clone() for k, p in m.named_parameters()})
    m.zero_grad(set_to_none=True)
    run(1.0)
    run(0.5)                                            # accumulates into the existing .grad tensors
    for k, p in m.named_parameters():
        want = single[0][k] + single[1][k]
        assert _rel(p.grad.cpu().numpy(), want.cpu().numpy()) <= 1e-5, k
    m.zero_grad(set_to_none=False)                      # in-place zero, then one more pass: must equal that pass alone
    run(1.0)
    for k, p in m.named_parameters():
        assert _rel(p.grad.cpu().numpy(), single[0][k].cpu().numpy()) <= 1e-5, k
