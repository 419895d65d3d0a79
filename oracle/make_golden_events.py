"""Record golden vectors for the event representation from the UNMODIFIED reference class
(/root/reference/data/utils/representations.py).  Run once in the build container:  python oracle/make_golden_events.py
Writes tests/golden/stacked_hist_<case>.npz (inputs + reference output)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
from data.utils.representations import StackedHistogram as RefHist          # noqa: E402
from tests.casegen_events import EVENT_CASES, make_events                   # noqa: E402

for name, c in EVENT_CASES.items():
    x, y, p, t = make_events(name)
    ref = RefHist(c['bins'], c['H'], c['W'], c['cutoff'], c['fastmode'])
    out = ref.construct(*(torch.from_numpy(a) for a in (x, y, p, t))).numpy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', f'stacked_hist_{name}.npz'), x=x, y=y, pol=p, time=t, out=out)
    print(name, out.shape, int(out.sum()), int(out.max()))
