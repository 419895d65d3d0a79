"""Binding of the C-side stage driver (include/rvt_hip.h: rvt_stage_seq_fwd) — the no-grad forward of one stage over a whole
sequence as ONE library call (validation / streaming inference, reference modules/detection.py:231-255).

rvt_amd/stage.py stays the host loop of the TRAINING forward / backward (it owns the saved-activation bookkeeping); this module
builds the descriptor records the library reads and calls it.  Python only allocates: the output slab, the final cell state and a
grow-only workspace per stream."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib as L

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


class RvtBlockWeights(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ('n1_w', 'n1_b', 'qkv_w', 'qkv_b', 'proj_w', 'proj_b', 'g1', 'n2_w', 'n2_b', 'fc1_w', 'fc1_b',
                                   'fc2_w', 'fc2_b', 'g2')]


class RvtStageDesc(ctypes.Structure):
    _fields_ = [('struct_bytes', _i), ('dtype', _i), ('C', _i), ('Cin', _i), ('cin_pad', _i), ('H_in', _i), ('W_in', _i), ('k', _i),
                ('stride', _i), ('pad', _i), ('ph', _i), ('pw', _i), ('dim_head', _i), ('num_blocks', _i), ('eps', _f),
                ('inp_u8', _i), ('h_raw', _i), ('w_raw', _i), ('conv_w', _vp), ('ln_w', _vp), ('ln_b', _vp),
                ('blocks', ctypes.POINTER(RvtBlockWeights)), ('lstm_w', _vp), ('lstm_b', _vp), ('lstm_wn', _vp), ('lstm_bn', _vp)]


def supported(sw, token_mask) -> bool:
    """What the driver covers: the 1x1-conv ConvLSTM of every shipped config, no token masks."""
    return token_mask is None and sw.dws is None


class StageCall:
    """Descriptor of one stage for one input flavour; keeps the host-side block array (and through `sw` the device tensors) alive."""

    def __init__(self, sw, g, dtype: torch.dtype, inp_u8: bool, h_raw: int, w_raw: int):
        p = L.ptr
        nb = 2 * g.num_blocks
        self.blocks = (RvtBlockWeights * max(nb, 1))()
        i = 0
        for pair in sw.blocks:
            for bw in pair:
                b = self.blocks[i]
                for k in ('n1_w', 'n1_b', 'qkv_w', 'qkv_b', 'proj_w', 'proj_b', 'g1', 'n2_w', 'n2_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'g2'):
                    setattr(b, k, p(bw[k]))
                i += 1
        d = RvtStageDesc()
        d.struct_bytes = ctypes.sizeof(RvtStageDesc)
        d.dtype, d.C, d.Cin, d.cin_pad = L.dtype_code(dtype), g.C, g.Cin, sw.cin_pad
        d.H_in, d.W_in, d.k, d.stride, d.pad = g.H_in, g.W_in, g.k, g.stride, g.pad
        d.ph, d.pw, d.dim_head, d.num_blocks, d.eps = g.ph, g.pw, g.dim_head, g.num_blocks, float(g.eps)
        d.inp_u8, d.h_raw, d.w_raw = int(inp_u8), h_raw, w_raw
        d.conv_w, d.ln_w, d.ln_b = p(sw.conv_w), p(sw.ln_w), p(sw.ln_b)
        d.blocks = ctypes.cast(self.blocks, ctypes.POINTER(RvtBlockWeights))
        d.lstm_w, d.lstm_b, d.lstm_wn, d.lstm_bn = p(sw.lstm_w), p(sw.lstm_b), p(sw.lstm_wn), p(sw.lstm_bn)
        self.desc, self.sw, self.g, self.dtype = d, sw, g, dtype
        self._ws_bytes = {}

    def ws_bytes(self, T: int, B: int) -> int:
        n = self._ws_bytes.get((T, B))
        if n is None:
            n = int(_lib().rvt_stage_seq_fwd_ws_bytes(ctypes.byref(self.desc), T, B))
            self._ws_bytes[(T, B)] = n
        return n


_bound = None


def _lib():
    global _bound
    lib = L.get_lib()
    if _bound is not lib:
        lib.rvt_stage_seq_fwd_ws_bytes.restype = ctypes.c_size_t
        lib.rvt_stage_seq_fwd_ws_bytes.argtypes = [_vp, _i, _i]
        lib.rvt_stage_seq_fwd.restype = _i
        lib.rvt_stage_seq_fwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _i, _i, _vp]
        _bound = lib
    return lib


_WS = {}


def stage_seq_fwd(call: StageCall, inp: torch.Tensor, h0: Optional[torch.Tensor], c0: Optional[torch.Tensor], T: int, B: int) \
        -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (Hall (T+1,B,H,W,C) in the compute dtype — slot 0 scratch, slots 1..T = h_t —, c_last (B,H,W,C) fp32)."""
    g, dt, dev = call.g, call.dtype, inp.device
    lib = _lib()
    st = L.stream_of(inp)
    n = call.ws_bytes(T, B)
    key = (dev.type, dev.index, 0 if st is None else int(st))
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        _WS[key] = ws
    Hall = torch.empty((T + 1, B, g.H, g.W, g.C), dtype=dt, device=dev)
    c_last = torch.empty((B, g.H, g.W, g.C), dtype=torch.float32, device=dev)
    if h0 is not None:
        assert h0.dtype == dt and h0.is_contiguous() and c0 is not None and c0.dtype == torch.float32 and c0.is_contiguous()
    rc = lib.rvt_stage_seq_fwd(ctypes.byref(call.desc), L.ptr(inp), L.ptr(h0), L.ptr(c0), L.ptr(Hall), L.ptr(c_last), L.ptr(ws),
                               ws.numel(), T, B, st)
    if rc != 0:
        raise RuntimeError(f'rvt_stage_seq_fwd failed: {lib.rvt_last_error().decode()}')
    return Hall, c_last
