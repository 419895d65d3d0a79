"""ctypes binding of librvt_hip.so (include/rvt_hip.h) — the only compute path of this package.

There is no PyTorch / CPU fallback: if the gfx950 library or a GPU is missing, every op raises.
The unit tests may install the CPU SIMT-emulator build of the same kernel sources through
``_install_test_library`` (tests/emu); that hook is never used by the package itself.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RVT_HIP_LIB') or os.path.join(_HERE, 'librvt_hip.so')   # env: another BUILD of the same library

RVT_F32, RVT_BF16 = 0, 1
_DT = {torch.float32: RVT_F32, torch.bfloat16: RVT_BF16}

_lib: Optional[ctypes.CDLL] = None
_is_emu = False

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_SIGS = {
    'rvt_stacked_histogram': [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    'rvt_prepack_input': [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_conv_fwd': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_conv_dgrad4': [_vp] * 4 + [_i] * 6 + [_vp],
    'rvt_conv_dgrad': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_conv_wgrad': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_stem_fwd': [_vp] * 6 + [_i] * 8 + [_f, _vp],
    'rvt_stem_wgrad': [_vp] * 4 + [_i] * 8 + [_vp],
    'rvt_layernorm_fwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'rvt_layernorm_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'rvt_linear_fwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'rvt_linear_scale_res_fwd': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'rvt_linear_dgrad': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'rvt_linear_dgrad_ln': [_vp] * 8 + [_i, _i, _i, _i, _f, _vp],
    'rvt_linear_dgrad_preln': [_vp] * 8 + [_i, _i, _i, _i, _f, _vp],
    'rvt_ln_linear_fwd': [_vp] * 7 + [_i, _i, _i, _i, _f, _vp],
    'rvt_linear_gelu_fwd': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'rvt_linear_wgrad': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'rvt_mlp_fwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'rvt_mlp_bwd_dgrad': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'rvt_attn_fwd': [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_attn_bwd': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_attn_block_fwd': [_vp] * 10 + [_i] * 9 + [_f, _vp],
    'rvt_attn_block_bwd': [_vp] * 12 + [_i] * 9 + [_f, _vp],
    'rvt_attn_block_bwd_preln': [_vp] * 11 + [_i] * 9 + [_f, _vp],
    'rvt_lstm_fwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    'rvt_lstm_gates_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    'rvt_lstm_dgrad': [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    'rvt_lstm_wgrad': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    'rvt_dwconv_fwd': [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_dwconv_wgrad': [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'rvt_token_mask_fwd': [_vp, _vp, _vp, _i, _i, _i, _vp],
    'rvt_token_mask_bwd': [_vp, _vp, _vp, _i, _i, _i, _vp],
    'rvt_state_reset_masked': [_vp, _vp, _i, _i, _sz, _vp],
    'rvt_gather_frames': [_vp, _vp, _vp, _i, _sz, _i, _vp],
    'rvt_pack_table': [_vp, _i, _i, _i, _vp],
    'rvt_mlp_bwd_recompute_dgrad': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'rvt_mlp_bwd_recompute_wgrad': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'rvt_mlp_bwd_recompute_both': [_vp] * 16 + [_i, _i, _i, _f, _vp],
    'rvt_lstm_scan_fwd': [_vp] * 8 + [_i, _i, _i, _i, _vp],
    'rvt_lstm_scan_bwd': [_vp] * 17 + [_i, _i, _i, _i, _vp],
    'rvt_lstm_scan3_pack': [_vp, _vp, _vp, _i, _vp],
    'rvt_lstm_scan3_fwd': [_vp] * 8 + [_i, _i, _i, _i, _vp],
    'rvt_lstm_scan3_bwd': [_vp] * 10 + [_i, _i, _i, _i, _vp],
    'rvt_layerscale_grad_table': [_vp, _i, _i, _vp],
    'rvt_bn_stats': [_vp, _vp, _vp, _i, _i, _i, _vp],
    'rvt_bn_finalize': [_vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    'rvt_bn_act_fwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'rvt_bn_train_act_fwd': [_vp, _vp, _vp, _i, _vp, _vp, _f, _f] + [_vp] * 7 + [_i, _i, _i, _i, _vp],
    'rvt_bn_act_bwd_stats': [_vp] * 8 + [_i, _i, _i, _i, _vp],
    'rvt_bn_act_bwd_apply': [_vp] * 9 + [_i, _i, _i, _i, _vp],
    'rvt_conv_bn_act_fwd': [_vp] * 5 + [_i] * 10 + [_vp],
    'rvt_yolox_decode': [_vp, _vp] + [_i] * 10 + [_vp, _vp, _vp],
    'rvt_yolox_decode_bwd': [_vp] * 5 + [_i] * 10 + [_vp],
    'rvt_simota_loss': [_vp] * 4 + [_i] * 5 + [_vp] * 5 + [ctypes.c_size_t, _vp],
}
EXPORTS = sorted(list(_SIGS) + ['rvt_last_error', 'rvt_is_emulator', 'rvt_wgrad_workspace_floats',
                               'rvt_mlp_fused_supported', 'rvt_lstm_scan_supported', 'rvt_mlp_bwd_fused_supported',
                               'rvt_mlp_bwd_fused_ws_floats', 'rvt_attn_block_supported', 'rvt_lstm_scan_bwd_ws_floats',
                               'rvt_lstm_scan_saves_gates', 'rvt_stem_supported', 'rvt_stem_wgrad_ws_floats', 'rvt_conv_dgrad4_supported',
                               'rvt_linear_dgrad_ln_supported', 'rvt_ln_linear_supported', 'rvt_tuning_defaults', 'rvt_get_tuning', 'rvt_set_tuning', 'rvt_probe_mfma',
                               'rvt_stage_seq_fwd', 'rvt_stage_seq_fwd_ws_bytes', 'rvt_lstm_scan3_supported', 'rvt_lstm_scan3_rows', 'rvt_stage_seq_train_fwd', 'rvt_stage_seq_bwd', 'rvt_stage_seq_bwd_ws_bytes', 'rvt_simota_ws_bytes', 'rvt_mlp_bwd_both_supported'])


def _bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.rvt_last_error.restype = ctypes.c_char_p
    lib.rvt_last_error.argtypes = []
    lib.rvt_is_emulator.restype = ctypes.c_int
    lib.rvt_is_emulator.argtypes = []
    lib.rvt_mlp_fused_supported.restype = ctypes.c_int
    lib.rvt_mlp_fused_supported.argtypes = [_i, _i]
    lib.rvt_mlp_bwd_fused_supported.restype = ctypes.c_int
    lib.rvt_mlp_bwd_fused_supported.argtypes = [_i, _i]
    lib.rvt_mlp_bwd_both_supported.restype = ctypes.c_int
    lib.rvt_mlp_bwd_both_supported.argtypes = [_i, _i]
    lib.rvt_mlp_bwd_fused_ws_floats.restype = ctypes.c_size_t
    lib.rvt_mlp_bwd_fused_ws_floats.argtypes = [_i, _i, _i]
    lib.rvt_lstm_scan_bwd_ws_floats.restype = ctypes.c_size_t
    lib.rvt_lstm_scan_bwd_ws_floats.argtypes = [_i, _i, _i]
    lib.rvt_lstm_scan_saves_gates.restype = ctypes.c_int
    lib.rvt_lstm_scan_saves_gates.argtypes = [_i, _i]
    lib.rvt_attn_block_supported.restype = ctypes.c_int
    lib.rvt_attn_block_supported.argtypes = [_i, _i, _i, _i]
    lib.rvt_lstm_scan_supported.restype = ctypes.c_int
    lib.rvt_lstm_scan_supported.argtypes = [_i, _i]
    lib.rvt_lstm_scan3_supported.restype = ctypes.c_int
    lib.rvt_lstm_scan3_supported.argtypes = [_i, _i]
    lib.rvt_lstm_scan3_rows.restype = ctypes.c_int
    lib.rvt_lstm_scan3_rows.argtypes = [_i, _i]
    lib.rvt_linear_dgrad_ln_supported.restype = ctypes.c_int
    lib.rvt_linear_dgrad_ln_supported.argtypes = [_i, _i, _i]
    lib.rvt_ln_linear_supported.restype = ctypes.c_int
    lib.rvt_ln_linear_supported.argtypes = [_i, _i, _i]
    lib.rvt_conv_dgrad4_supported.restype = ctypes.c_int
    lib.rvt_conv_dgrad4_supported.argtypes = [_i] * 9
    lib.rvt_stem_supported.restype = ctypes.c_int
    lib.rvt_stem_supported.argtypes = [_i] * 8
    lib.rvt_stem_wgrad_ws_floats.restype = ctypes.c_size_t
    lib.rvt_stem_wgrad_ws_floats.argtypes = [_i] * 4
    lib.rvt_wgrad_workspace_floats.restype = ctypes.c_size_t
    lib.rvt_wgrad_workspace_floats.argtypes = [_i, _i, _i, _i, _i]
    lib.rvt_simota_ws_bytes.restype = ctypes.c_size_t
    lib.rvt_simota_ws_bytes.argtypes = [_i, _i, _i]
    lib.rvt_probe_mfma.restype = ctypes.c_double
    lib.rvt_probe_mfma.argtypes = [_vp, _i, _i, _vp]
    lib.rvt_tuning_defaults.restype = None
    lib.rvt_tuning_defaults.argtypes = [_vp]
    lib.rvt_get_tuning.restype = ctypes.c_int
    lib.rvt_get_tuning.argtypes = [_vp]
    lib.rvt_set_tuning.restype = ctypes.c_int
    lib.rvt_set_tuning.argtypes = [_vp]
    return lib


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen + bind every symbol of include/rvt_hip.h (no GPU needed to *load*)."""
    if not os.path.exists(path):
        raise RuntimeError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'(rvt_amd/csrc/build.sh).  rvt_amd has no fallback compute path.')
    return _bind(ctypes.CDLL(path))


def get_lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError('rvt_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False '
                               'and there is no CPU fallback.')
        _lib = lib
        from . import tuning
        tuning.push(lib)        # the caller's overrides (none = the production route) into the freshly loaded library
    return _lib


def _install_test_library(lib: Optional[ctypes.CDLL]) -> None:
    """TEST HOOK ONLY (tests/emu): route calls to the CPU SIMT-emulator build of the kernel sources."""
    global _lib, _is_emu
    _lib = lib
    _is_emu = bool(lib is not None and lib.rvt_is_emulator())
    if lib is not None:
        from . import tuning
        tuning.push(lib)


def is_emulator() -> bool:
    return _is_emu


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f'rvt_amd kernels support float32 and bfloat16 activations, got {dt}') from None


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError('rvt_amd kernels need contiguous tensors')
    if not (t.is_cuda or _is_emu):
        raise RuntimeError('rvt_amd kernels need CUDA (ROCm) tensors; there is no CPU path')
    return t.data_ptr()


class _Stream(int):
    """hipStream_t handle (as an int for ctypes) that remembers the device it belongs to."""
    dev = -1


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_of(t: torch.Tensor) -> Optional[int]:
    """Current HIP stream of the tensor's device.  Every entry point takes it as its LAST argument; call() uses the
    device it carries to make that device current for the launch (the C ABI has no device parameter: a kernel launched
    on stream S runs on S's device, but occupancy queries and the null stream follow the thread's current device)."""
    if not t.is_cuda:
        return None
    idx = t.device.index
    s = _Stream(_raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(t.device).cuda_stream)
    s.dev = idx
    return s


def call(name: str, *args) -> None:
    lib = get_lib()
    st = args[-1] if args else None
    if isinstance(st, _Stream) and st.dev != torch.cuda.current_device():
        with torch.cuda.device(st.dev):
            rc = getattr(lib, name)(*args)
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed: {lib.rvt_last_error().decode()}')
