#!/bin/bash
# Build librvt_hip.so (gfx950 code object + host launchers) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../librvt_hip.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result \
    -ffp-contract=off -fno-honor-nans capi.hip -o "$OUT" "$@"
echo "built $(realpath $OUT)"
