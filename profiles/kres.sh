#!/bin/bash
# register / LDS / scratch table of the kernels of ONE translation unit: profiles/kres.sh <part> [filter-substring]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/rvt_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -ffp-contract=off -fno-honor-nans \
  -Rpass-analysis=kernel-resource-usage -c capi_$1.hip -o /tmp/kres_$1.o 2>&1 | python3 $ROOT/profiles/kernel_resources.py "$2"
