#!/bin/bash
# Build the CPU SIMT-emulator flavour of the SAME kernel sources (tests only; see hip_emu.h).
set -e
cd "$(dirname "$0")"
CXX=/opt/rocm/lib/llvm/bin/clang++
OUT=librvt_emu.so
$CXX -x c++ -std=c++17 -O2 -fPIC -shared -DRVT_EMU -I. -include hip_emu.h -Wno-unknown-attributes \
    -ffp-contract=off ../../rvt_amd/csrc/capi.hip -x c++ emu.cpp -o "$OUT"
echo "built $(realpath $OUT)"
