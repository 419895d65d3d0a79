"""Event stream -> stacked histogram: the HIP kernel vs the numpy oracle and vs golden vectors
recorded from the reference class.  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle.events_oracle import stacked_histogram as oracle_hist
from rvt_amd.representations import StackedHistogram
from tests.backends import backend  # noqa: F401
from tests.casegen_events import EVENT_CASES, make_events

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('name', list(EVENT_CASES))
def test_oracle_matches_reference_golden(name):
    c = EVENT_CASES[name]
    g = np.load(os.path.join(GOLD, f'stacked_hist_{name}.npz'))
    x, y, p, t = make_events(name)
    for a, b in ((x, g['x']), (y, g['y']), (p, g['pol']), (t, g['time'])):
        assert np.array_equal(a, b)                                  # the generator is what the recorder used
    out = oracle_hist(x, y, p, t, c['bins'], c['H'], c['W'], c['cutoff'], c['fastmode'])
    assert out.dtype == np.uint8 and np.array_equal(out, g['out'])


@pytest.mark.parametrize('name', list(EVENT_CASES))
def test_hip_matches_golden_and_oracle(backend, name):
    c = EVENT_CASES[name]
    g = np.load(os.path.join(GOLD, f'stacked_hist_{name}.npz'))
    x, y, p, t = (torch.from_numpy(a).to(backend) for a in make_events(name))
    rep = StackedHistogram(c['bins'], c['H'], c['W'], c['cutoff'], c['fastmode'])
    out = rep.construct(x, y, p, t)
    assert out.dtype == torch.uint8 and tuple(out.shape) == rep.get_shape()
    assert np.array_equal(out.cpu().numpy(), g['out'])
    # a second call reuses the scratch image: must not accumulate across calls
    assert np.array_equal(rep.construct(x, y, p, t).cpu().numpy(), g['out'])
    # int32 inputs (the reference accepts any integer dtype)
    out32 = rep.construct(x.to(torch.int32), y.to(torch.int32), p.to(torch.int32), t)
    assert np.array_equal(out32.cpu().numpy(), g['out'])


@pytest.mark.gpu
def test_full_size_1mpx_histogram_properties():
    """20 M events on the 1 Mpx sensor (no oracle run at this size): total count == events when nothing saturates, and the
    result does not depend on how the stream is chunked in time-sorted order (count additivity before the clamp)."""
    H, W, bins, n = 720, 1280, 10, 20_000_000
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randint(0, W, (n,), generator=g, device='cuda')
    y = torch.randint(0, H, (n,), generator=g, device='cuda')
    p = torch.randint(0, 2, (n,), generator=g, device='cuda')
    t = torch.sort(torch.randint(0, 50_000, (n,), generator=g, device='cuda')).values
    rep = StackedHistogram(bins, H, W, count_cutoff=None, fastmode=False)
    out = rep.construct(x, y, p, t)
    assert int(out.sum(dtype=torch.int64)) == n                      # ~1 event per cell: nothing reaches 255
    # per-bin totals agree with a direct count of the float32 bin rule
    ti = torch.clamp(((t - t[0]).float() / float(max(int(t[-1] - t[0]), 1)) * bins).floor(), max=bins - 1).long()
    per_bin = out.view(2, bins, H, W).sum(dim=(0, 2, 3), dtype=torch.int64)
    assert torch.equal(per_bin, torch.bincount(ti, minlength=bins))
