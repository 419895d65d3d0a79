"""bf16 performance mode against the fp32 HIP path over the FULL recurrent depth of the benchmarked configuration
(RVT-Base, 1 Mpx, T=21; B=1 to keep it cheap): per-time-step drift of every stage's features, the final cell states and
the parameter gradients.  The golden comparisons (test_backbone.py) stop at T<=5; this one measures what 21 recurrent
steps of 8-mantissa-bit storage do and pins a bound on it.  GPU only."""
import pytest
import torch

from rvt_amd import RNNDetector, backbone_config

pytestmark = pytest.mark.gpu

# measured on MI355X (profiles/r2_drift.txt): worst per-step feature drift 2.1e-2 of the tensor scale (stages 3-4), flat in
# t after the first few steps (the LSTM gates are contractive: no growth over 21 recurrent steps); final cell states 9.5e-3;
# parameter gradients 1.35e-2 of their l2 norm.  Bounds = ~2x the measurement.
FEATURE_BOUND = 4e-2
GRAD_BOUND = 3e-2


def test_bf16_drift_over_21_steps():
    dev = torch.device('cuda', 0)
    T, B = 21, 1
    g = torch.Generator().manual_seed(1)
    xs = torch.randint(0, 11, (T, B, 20, 360, 640), generator=g, dtype=torch.uint8).to(dev)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        m = RNNDetector(backbone_config('base', 'gen4'), compute_dtype=dt).to(dev)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith('gamma'):
                    p.uniform_(0.5, 1.0, generator=torch.Generator(device=dev).manual_seed(5))
        geoms = m.stage_geoms(384, 640)
        gc = torch.Generator(device=dev).manual_seed(7)
        cots = [torch.randn((T, B, geoms[s].C, geoms[s].H, geoms[s].W), device=dev, generator=gc) for s in range(4)]
        feats, states = m.forward_sequence(xs)
        torch.autograd.backward([feats[s + 1] for s in range(4)], [c.to(feats[s + 1].dtype) for s, c in enumerate(cots)])
        res[dt] = ({s: feats[s].detach().float() for s in feats}, [c.detach().float() for _, c in states],
                   {n: p.grad.detach().float().clone() for n, p in m.named_parameters()})
        del m, feats, states
    f32, b16 = res[torch.float32], res[torch.bfloat16]
    lines = []
    worst_f = 0.0
    for s in range(1, 5):
        scale = float(f32[0][s].abs().max())
        per_t = [float((b16[0][s][t] - f32[0][s][t]).abs().max()) / scale for t in range(T)]
        worst_f = max(worst_f, max(per_t))
        lines.append(f'stage {s} feature drift / scale per t: ' + ' '.join(f'{v:.1e}' for v in per_t))
    for s in range(4):
        scale = float(f32[1][s].abs().max())
        d = float((b16[1][s] - f32[1][s]).abs().max()) / scale
        worst_f = max(worst_f, d)
        lines.append(f'stage {s + 1} final cell state drift / scale: {d:.2e}')
    worst_g = 0.0
    for n in f32[2]:
        nrm = float(f32[2][n].norm())
        d = float((b16[2][n] - f32[2][n]).norm()) / max(nrm, 1e-30)
        worst_g = max(worst_g, d)
    lines.append(f'worst parameter-gradient l2 drift: {worst_g:.2e}; worst feature/state drift: {worst_f:.2e}')
    print('\n'.join(lines))
    import os
    out = os.environ.get('RVT_DRIFT_REPORT')
    if out:
        with open(out, 'w') as f:
            f.write('\n'.join(lines) + '\n')
    assert worst_f <= FEATURE_BOUND, worst_f
    assert worst_g <= GRAD_BOUND, worst_g
