"""Seeded event streams for the stacked-histogram tests (shared by the golden recorder and the tests)."""
import numpy as np

EVENT_CASES = {
    # name: geometry, event count, options
    'small': dict(bins=10, H=24, W=32, n=5000, cutoff=10, fastmode=True, span=50000, hot=0),
    'no_cutoff_int16': dict(bins=4, H=16, W=16, n=3000, cutoff=None, fastmode=False, span=1000, hot=400),
    'hot_pixel_wrap': dict(bins=2, H=8, W=8, n=2000, cutoff=None, fastmode=True, span=100, hot=700),   # > 255 at one cell
    'single_timestamp': dict(bins=10, H=12, W=20, n=300, cutoff=10, fastmode=True, span=0, hot=0),
    'empty': dict(bins=10, H=12, W=20, n=0, cutoff=10, fastmode=True, span=0, hot=0),
    'gen1_like': dict(bins=10, H=240, W=304, n=200000, cutoff=10, fastmode=True, span=49999, hot=0),
}


def make_events(name):
    c = EVENT_CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    n = c['n']
    x = rng.integers(0, c['W'], n, dtype=np.int64)
    y = rng.integers(0, c['H'], n, dtype=np.int64)
    p = rng.integers(0, 2, n, dtype=np.int64)
    t = np.sort(rng.integers(0, c['span'] + 1, n, dtype=np.int64)) + 1_000_000
    if c['hot'] and n:
        k = min(c['hot'], n)                       # a hot pixel: many events in one cell (accumulator wrap-around)
        x[:k], y[:k], p[:k] = 3, 5, 1
        t[:k] = t[0]
    return x, y, p, t
