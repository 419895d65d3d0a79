"""hipGraph capture through the C ABI (include/rvt_hip.h promises: asynchronous, allocate nothing, graph-capturable) and
of the whole training step (rvt_amd/graph.py — the reference's analogue is torch.compile(mode='reduce-overhead'),
maxvit_rnn.py:43-51).  GPU only."""
import pytest
import torch

from rvt_amd import ops

pytestmark = pytest.mark.gpu


def _chain(x, w, b, lw, lb, wt, out_y, out_dx):
    """A few C-ABI launches of different kernel families writing into preallocated outputs."""
    u = ops.layernorm_fwd(x, lw, lb, 1e-5)
    y = ops.linear_fwd(u, w, b, out=out_y)
    ops.linear_dgrad(y, wt, out=out_dx)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_c_abi_launches_capture_and_replay(dt):
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(0)
    M, K, N = 4096, 64, 192
    x = torch.randn(M, K, device=dev, generator=g).to(dt)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.2).to(dt)
    wt = w.t().contiguous()
    b = torch.randn(N, device=dev, generator=g)
    lw, lb = torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.1
    y, dx = torch.empty(M, N, device=dev, dtype=dt), torch.empty(M, K, device=dev, dtype=dt)
    _chain(x, w, b, lw, lb, wt, y, dx)                       # eager warm-up (occupancy caches, allocator)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _chain(x, w, b, lw, lb, wt, y, dx)
    for seed in (1, 2):                                      # new input CONTENTS at the same addresses
        x.copy_(torch.randn(M, K, device=dev, generator=g.manual_seed(seed)).to(dt))
        y.zero_(); dx.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got_y, got_dx = y.clone(), dx.clone()
        ey, edx = torch.empty_like(y), torch.empty_like(dx)
        _chain(x, w, b, lw, lb, wt, ey, edx)
        torch.cuda.synchronize()
        assert torch.equal(got_y, ey) and torch.equal(got_dx, edx)


def test_training_step_graph_matches_eager():
    """Whole step (forward_sequence + BPTT backward with the weight-gradient side stream + fused AdamW) captured once and
    replayed: parameters after 3 replays == parameters after 3 eager steps, bit for bit (same kernels, same order)."""
    from rvt_amd import RNNDetector, backbone_config
    from rvt_amd.graph import GraphedStep
    dev = torch.device('cuda', 0)

    def make():
        torch.manual_seed(0)
        m = RNNDetector(backbone_config('tiny', 'gen1'), compute_dtype=torch.bfloat16).to(dev)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith('gamma'):
                    p.fill_(0.7)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)
        return m, opt
    T, B = 3, 2
    g = torch.Generator(device=dev).manual_seed(1)
    xs = torch.randint(0, 11, (T, B, 20, 240, 304), generator=g, dtype=torch.uint8, device=dev)
    geoms = make()[0].stage_geoms(256, 320)
    cots = [torch.randn((T, B, geoms[s].C, geoms[s].H, geoms[s].W), device=dev, generator=g).to(torch.bfloat16) for s in range(4)]

    def step_of(m, opt):
        def step():
            feats, _ = m.forward_sequence(xs, None)
            torch.autograd.backward([feats[s + 1] for s in range(4)], cots)
            opt.step()
            opt.zero_grad(set_to_none=True)
        return step
    m_e, opt_e = make()
    m_g, opt_g = make()
    init = {n: p.detach().clone() for n, p in m_e.named_parameters()}
    eager, graphed_fn = step_of(m_e, opt_e), step_of(m_g, opt_g)
    warm = 2
    gs = GraphedStep(graphed_fn, warmup=warm, models=(m_g,))      # the warm-up steps are REAL (eager) steps; the capture pass only records
    for _ in range(warm):
        eager()
    for _ in range(3):
        gs()
        eager()
    torch.cuda.synchronize()
    # same kernels in the same order; the only run-to-run freedom is the order of the fp32 atomics that fold the LayerNorm
    # parameter gradients across workgroups, so compare against how far the parameters moved at all
    num = den = 0.0
    for (n, a), (_, b) in zip(m_g.named_parameters(), m_e.named_parameters()):
        num += float((a - b).double().pow(2).sum())
        den += float((b - init[n]).double().pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < 2e-2, (num, den)
    # a replay moves the parameters without touching their host-side version counters: the next inference forward must re-pack
    # the kernel-side weights (GraphedStep invalidates the cache), i.e. agree with the eager model on the same parameters
    with torch.no_grad():
        fg, _ = m_g.forward_sequence(xs, None)
        for (_, a), (_, b) in zip(m_e.named_parameters(), m_g.named_parameters()):
            a.copy_(b)
        m_e.invalidate_weight_cache()
        fe, _ = m_e.forward_sequence(xs, None)
    for s_ in range(1, 5):
        assert torch.equal(fg[s_], fe[s_])


def test_streaming_step_graph_matches_eager():
    """Streaming inference (T = 1, state carried across calls; modules/detection.py:231-255) replayed as one hipGraph
    (rvt_amd.graph.GraphedStreamStep): features of every step == the eager loop's, bit for bit, incl. a masked state reset."""
    from rvt_amd import RNNDetector, backbone_config
    from rvt_amd.graph import GraphedStreamStep
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    m = RNNDetector(backbone_config('tiny', 'gen1'), compute_dtype=torch.bfloat16).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    frames = [torch.randint(0, 11, (3, 20, 240, 304), generator=g, dtype=torch.uint8, device=dev) for _ in range(5)]
    mask = torch.tensor([False, True, False], device=dev)
    want = []
    with torch.no_grad():
        st = None
        for i, f in enumerate(frames):
            if i == 3:                                         # sequence boundary of sample 1: its state restarts from zeros
                st = [(h.clone(), c.clone()) for h, c in st]
                for h, c in st:
                    h[mask] = 0
                    c[mask] = 0
            feats, st = m(f, st)
            want.append({k: v.clone() for k, v in feats.items()})
    gs = GraphedStreamStep(m, frames[0])
    for i, f in enumerate(frames):
        if i == 3:
            gs.reset(mask)
        feats = gs(f)
        torch.cuda.synchronize()
        for k in want[i]:
            assert torch.equal(feats[k], want[i][k]), (i, k)
    gs.close()
