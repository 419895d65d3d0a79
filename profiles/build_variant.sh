#!/bin/bash
# Build the library as it was at a git revision into rvt_amd/librvt_hip_<suffix>.so (same-box A/B: RVT_HIP_LIB=<that file> selects it).
#   usage: bash profiles/build_variant.sh <git-rev> <suffix>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}; SUF=${2:-base}
D=/tmp/rvt_variant_$SUF
rm -rf $D; mkdir -p $D/rvt_amd $D/include
git -C $ROOT archive $REV rvt_amd/csrc include | tar -x -C $D
make -C $D/rvt_amd/csrc -j8 OUT=$ROOT/rvt_amd/librvt_hip_$SUF.so 2>&1 | grep -v "warning\|^ \|^$\|generated" | tail -2
ls -la $ROOT/rvt_amd/librvt_hip_$SUF.so
