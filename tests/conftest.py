import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Test-size problems have only a handful of GEMM tiles, so the unit tests run on an explicit TEST GEOMETRY
# (rvt_amd/tuning.py: TEST_GEOMETRY — 8-workgroup persistent grids so that gemm_kernel / ppgemm walk several tiles per
# workgroup, 128-token split-K slices so that the two-stage reduction runs, the ConvLSTM scan and the 256 x 256 LDS-DMA GEMM
# family at every width they are built for, 3-workgroup grids for the one-per-CU kernels).  It is installed through the C
# ABI (rvt_set_tuning), not through the environment, and it is NOT what bench.py times: the production route (library
# defaults) is covered by tests/test_production_route.py, which switches to `tuning.production()` explicitly.
from rvt_amd import tuning  # noqa: E402

tuning.use(**tuning.TEST_GEOMETRY)


@pytest.fixture
def production_route():
    """Run the test body on the library defaults — the launch geometry and kernel routing bench.py times."""
    saved = tuning.overrides()
    tuning.production()
    try:
        yield
    finally:
        tuning.use(**saved)


from tests.cores import usable_cores  # noqa: E402,F401


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
