"""Size-independent properties of the HIP path at the BASELINE.json resolution (RVT-Base, 1 Mpx, bf16 and fp32).

The golden / oracle comparisons in test_backbone.py run at sizes the CPU oracle finishes in seconds.  At the real
resolution (20x360x640 event tensors, padded to 384x640) the same code paths run with millions of token rows per
kernel (persistent multi-tile walks, several K slices per weight gradient), and what can be checked there without an
oracle are invariances the recurrent backbone has by construction:

  * a sequence processed in one call == the same sequence processed in two calls with the LSTM states carried over
    (reference semantics of modules/detection.py:131-148 + RNNStates: the time loop is a left fold);
  * sample b of a batch does not depend on the other samples (no cross-sample op in the backbone);
  * the backward pass is linear in the cotangents and the parameter gradients of a batch are the sum over samples.
"""
import pytest
import torch

from rvt_amd import RNNDetector, backbone_config

pytestmark = pytest.mark.gpu

HW = (360, 640)


def _model(dtype, seed=0):
    torch.manual_seed(seed)
    m = RNNDetector(backbone_config('base', 'gen4'), compute_dtype=dtype).cuda()
    with torch.no_grad():                       # LayerScale at its 1e-5 init would hide the residual branches
        for n, p in m.named_parameters():
            if n.endswith('gamma'):
                p.uniform_(0.5, 1.0)
    return m


def _inputs(T, B, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 11, (T, B, 20, *HW), generator=g, dtype=torch.uint8).cuda()


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_sequence_split_and_batch_independence(dtype):
    m = _model(dtype)
    T, B = 4, 2
    xs = _inputs(T, B)
    with torch.no_grad():
        feats, states = m.forward_sequence(xs)
        f1, s1 = m.forward_sequence(xs[:1])
        f2, s2 = m.forward_sequence(xs[1:], s1)
        fb, sb = m.forward_sequence(xs[:, 1:2])
    for s in range(1, 5):
        # every output row is produced by the same instruction sequence over the same operands whatever the launch
        # size: the forward is BIT-identical under both re-groupings
        assert torch.equal(feats[s][:1], f1[s]), s
        assert torch.equal(feats[s][1:], f2[s]), s
        assert torch.equal(feats[s][:, 1:2], fb[s]), s
        assert torch.equal(states[s - 1][1], s2[s - 1][1]), s          # cell states (fp32)
        assert torch.equal(states[s - 1][1][1:2], sb[s - 1][1]), s
        assert torch.isfinite(feats[s].float()).all()


def _grads(m, xs, cots, prev=None):
    for p in m.parameters():
        p.grad = None
    feats, states = m.forward_sequence(xs, prev)
    loss = sum((feats[s].float() * cots[s - 1]).sum() for s in range(1, 5))
    loss.backward()
    return {n: p.grad.clone() for n, p in m.named_parameters()}, feats


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_backward_linear_in_cotangents_and_additive_over_samples(dtype, tol):
    m = _model(dtype)
    T, B = 3, 2
    xs = _inputs(T, B, seed=2)
    with torch.no_grad():
        feats, _ = m.forward_sequence(xs)
    g = torch.Generator(device='cuda').manual_seed(3)
    c1 = [torch.randn(feats[s].shape, device='cuda', generator=g) for s in range(1, 5)]
    c2 = [torch.randn(feats[s].shape, device='cuda', generator=g) for s in range(1, 5)]
    g1, _ = _grads(m, xs, c1)
    g2, _ = _grads(m, xs, c2)
    g12, _ = _grads(m, xs, [a + 2.0 * b for a, b in zip(c1, c2)])
    # per-sample runs: gradients of the batch = sum of the gradients of its samples
    ga, _ = _grads(m, xs[:, :1], [c[:, :1] for c in c1])
    gb, _ = _grads(m, xs[:, 1:], [c[:, 1:] for c in c1])
    worst_lin, worst_add = 0.0, 0.0
    for n in g1:
        scale = g12[n].abs().max().clamp_min(1e-20)
        worst_lin = max(worst_lin, float(((g1[n] + 2.0 * g2[n]) - g12[n]).abs().max() / scale))
        scale = g1[n].abs().max().clamp_min(1e-20)
        worst_add = max(worst_add, float(((ga[n] + gb[n]) - g1[n]).abs().max() / scale))
        assert torch.isfinite(g1[n]).all(), n
    assert worst_lin <= tol, worst_lin
    assert worst_add <= tol, worst_add


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_truncated_bptt_gradients_flow_through_carried_states(dtype, tol):
    """d loss / d params of a T=4 sequence == the same loss evaluated as two T=2 calls whose states are NOT detached."""
    m = _model(dtype)
    xs = _inputs(4, 1, seed=4)
    with torch.no_grad():
        feats, _ = m.forward_sequence(xs)
    g = torch.Generator(device='cuda').manual_seed(5)
    cots = [torch.randn(feats[s].shape, device='cuda', generator=g) for s in range(1, 5)]
    whole, _ = _grads(m, xs, cots)
    for p in m.parameters():
        p.grad = None
    fa, sa = m.forward_sequence(xs[:2])
    fb, _ = m.forward_sequence(xs[2:], sa)
    loss = sum((fa[s].float() * cots[s - 1][:2]).sum() + (fb[s].float() * cots[s - 1][2:]).sum() for s in range(1, 5))
    loss.backward()
    worst = 0.0
    for n, p in m.named_parameters():
        scale = whole[n].abs().max().clamp_min(1e-20)
        worst = max(worst, float((p.grad - whole[n]).abs().max() / scale))
    assert worst <= tol, worst


def test_stem_route_equals_gemm_route_at_full_resolution():
    """The stem kernels on the uint8 planes (csrc/stem.hpp) against the prepack + im2col-GEMM + LayerNorm route they replace, at
    the bench resolution (interior fast paths, five 32-pixel segments per row, bottom zero padding 360 -> 384): the products
    are the same (uint8 is exact in bf16), only the fp32 summation order differs."""
    from rvt_amd import ops, weights
    dt = torch.bfloat16
    F_, Cin, (h, w), H, W = 3, 20, HW, 384, 640
    src = _inputs(1, F_, seed=3)[0].contiguous()
    g = torch.Generator().manual_seed(4)
    wt = (torch.randn(64, Cin, 7, 7, generator=g) * 0.05).cuda()
    wp = weights.pack_conv_fwd(wt, 24, dt)
    lw, lb = (1 + 0.2 * torch.randn(64, generator=g)).cuda(), (0.2 * torch.randn(64, generator=g)).cuda()
    assert ops.stem_supported(src, dt, 64, 7, 4, 3)
    y0, x = ops.stem_fwd(src, wp, lw, lb, H, W, 1e-5)
    inp = ops.prepack_input(src, H, W, 24, dt)
    y_ref = ops.conv_fwd(inp, wp, 7, 4, 3)
    x_ref = ops.layernorm_fwd(y_ref, lw, lb, 1e-5)
    assert _rel(y0, y_ref) < 8e-3 and _rel(x, x_ref) < 2e-2          # one bf16 ulp of the largest value / of a normalised row
    assert float((y0.float() - y_ref.float()).abs().mean()) < 2e-3 * float(y_ref.float().abs().mean())
    dy = (torch.randn(y0.shape, generator=g) * 0.5).to(dt).cuda()
    dw_a, dw_b = torch.zeros(64, 49 * 24).cuda(), torch.zeros(64, 49 * 24).cuda()
    ops.conv_wgrad(inp, dy, dw_a, 7, 4, 3)
    ops.stem_wgrad(src, dy, dw_b, H, W)
    assert _rel(dw_b, dw_a) < 1e-4                                    # fp32 accumulation of identical bf16 products
