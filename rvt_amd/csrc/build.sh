#!/bin/bash
# Build librvt_hip.so (gfx950 code objects + host launchers) in-tree: `make -j` over the ten capi_*.hip parts.
# A failed compile fails the script (pipefail: the grep only drops empty lines) — a stale library never passes for a fresh one.
set -e -o pipefail
cd "$(dirname "$0")"
make -j"$(nproc)" "$@" 2>&1 | { grep -v "^$" || true; }
test -f ../librvt_hip.so
test ! ../librvt_hip.so -ot _obj/capi_core.o
echo "built $(realpath ../librvt_hip.so)"
