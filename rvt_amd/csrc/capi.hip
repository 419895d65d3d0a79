// Unity translation unit of the C ABI (include/rvt_hip.h): the CPU SIMT-emulator build (tests/emu/build_emu.sh) compiles
// this one file; the gfx950 build (build.sh) compiles the ten capi_*.hip parts in parallel and links them.
#include "capi_core.hip"
#include "capi_conv.hip"
#include "capi_linear.hip"
#include "capi_stem.hip"
#include "capi_mlp.hip"
#include "capi_attn.hip"
#include "capi_lstm.hip"
#include "capi_scan.hip"
#include "capi_stage.hip"
#include "capi_head.hip"
#include "capi_train.hip"
