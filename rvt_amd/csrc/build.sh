#!/bin/bash
# Build librvt_hip.so (gfx950 code objects + host launchers) in-tree: `make -j` over the eight capi_*.hip parts.
set -e
cd "$(dirname "$0")"
make -j"$(nproc)" "$@" 2>&1 | grep -v "^$" || true
test -f ../librvt_hip.so
echo "built $(realpath ../librvt_hip.so)"
