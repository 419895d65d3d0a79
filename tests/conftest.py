import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Test-size problems have only a handful of GEMM tiles; shrink the persistent grid (normally ~2-3 workgroups per CU)
# so that the multi-tile walk + cross-tile prefetch of gemm_kernel is what the tests execute (emulator AND GPU).
os.environ.setdefault('RVT_GEMM_RESIDENT', '8')     # (a multiple of 8: also reaches the XCD-contiguous walk of the conv sources)
# ... and let the weight-gradient kernels cut even test-size token counts into several K slices (production: >= 8192
# tokens per slice), so the two-stage split-K path (partial tiles + reduction, column sums per slice) is exercised
os.environ.setdefault('RVT_WGRAD_SLICE_TOKENS', '128')
# ConvLSTM scan kernels for every width they are built for (production default: only where the weights fit the LDS)
os.environ.setdefault('RVT_LSTM_SCAN', '1')
# stem kernels (one workgroup per CU in production): a 3-workgroup grid, so that test sizes walk several items / tiles per
# workgroup (the double-buffered LDS image of the weight gradient, the persistent item loop of the forward)
os.environ.setdefault('RVT_STEM_GRID', '3')
# 256 x 256 LDS-DMA GEMM (csrc/ppgemm.hpp; production: >= 4096 rows, one workgroup per CU): test-size row counts, and an
# 8-workgroup grid so that a workgroup walks several output tiles (load stream / accumulator flush across tile boundaries)
os.environ.setdefault('RVT_PPGEMM_MIN_M', '256')
os.environ.setdefault('RVT_PPGEMM_GRID', '8')
os.environ.setdefault('RVT_PPGEMM_TN_ITEMS', '6')          # weight-gradient variant: a few token slices per output tile
os.environ.setdefault('RVT_PPGEMM_ALL', '1')            # every epilogue flavour through it, whatever the contraction length
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cores() -> int:
    """Cores this process may really use (affinity mask, capped by the cgroup quota and 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    import torch
    torch.set_num_threads(usable_cores())      # CPU oracle / emulator tests: never oversubscribe a throttled container
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
