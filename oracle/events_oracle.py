"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference event representation
``StackedHistogram.construct`` (data/utils/representations.py:76-117).  Pinned by tests/golden/stacked_hist_*.npz,
which oracle/make_golden_events.py records from the unmodified reference class.  Never imported by rvt_amd/."""
import numpy as np


def stacked_histogram(x, y, pol, time, bins, height, width, count_cutoff=None, fastmode=True):
    """x, y, pol, time: integer arrays [n] (time ascending).  Returns uint8 (2*bins, height, width)."""
    cutoff = 255 if count_cutoff is None else min(count_cutoff, 255)               # representations.py:52-57
    acc_dtype = np.uint8 if fastmode else np.int16                                  # :83
    rep = np.zeros(2 * bins * height * width, dtype=np.int64)
    if len(x) > 0:
        t = np.asarray(time, dtype=np.int64)
        t0, t1 = t[0], t[-1]                                                        # :99-101 (sorted time)
        den = np.float32(max(int(t1 - t0), 1))
        t_norm = (t - t0).astype(np.float32) / den                                  # :102-103, float32 like torch
        t_norm = t_norm * np.float32(bins)                                          # :104
        t_idx = np.minimum(np.floor(t_norm), bins - 1).astype(np.int64)             # :105-106
        idx = (np.asarray(x, np.int64) + width * np.asarray(y, np.int64) + height * width * t_idx
               + bins * height * width * np.asarray(pol, np.int64))                # :108-111
        np.add.at(rep, idx, 1)                                                      # :113 put_(accumulate=True)
    rep = rep.astype(acc_dtype)                                 # the accumulator wraps like the reference's dtype
    rep = np.clip(rep.astype(np.int64), 0, cutoff).astype(np.uint8)                 # :114-116
    return rep.reshape(2 * bins, height, width)                                     # :117 merge_channel_and_bins
