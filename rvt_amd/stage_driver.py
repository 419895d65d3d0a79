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


# ---- training-side driver (round 6; include/rvt_hip.h: rvt_stage_seq_train_fwd / rvt_stage_seq_bwd, csrc/capi_train.hip) ----------------
# The host decides the routes (the predicates of rvt_amd/stage.py - ONE place) and owns every tensor that outlives a call; the
# library sequences the launches.  `StageSaved` is filled exactly as the Python host loop fills it, so either backward can consume it.
class RvtBlockSaved(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ('xin', 'u', 'qkv', 'a', 'xmid', 'v2', 'hg', 'hgp', 'xout')]


class RvtBlockTrain(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ('qkv_wt', 'proj_wt', 'fc1_wt', 'fc2_wt', 'd_n1_w', 'd_n1_b', 'd_qkv_w', 'd_qkv_b', 'd_S1', 'd_cs1',
                                   'd_n2_w', 'd_n2_b', 'd_fc1_w', 'd_fc1_b', 'd_S2', 'd_cs2')]


class RvtStageTrain(ctypes.Structure):
    _fields_ = [('struct_bytes', _i)] + [(n, _i) for n in ('attn_block', 'ln_linear', 'mlp_route', 'mlp_bwd_both', 'dgrad_ln_qkv',
                                                          'dgrad_ln_fc1', 'lstm_route', 'lstm_scan_wgrad', 'conv_dgrad4', 'attn_preln')] + \
               [('saved', ctypes.POINTER(RvtBlockSaved)), ('tb', ctypes.POINTER(RvtBlockTrain))] + \
               [(n, _vp) for n in ('y0', 'x0', 'Hall', 'c_last', 'Csave', 'gates', 'Call', 'c0_saved', 'lstm_wp3', 'lstm_wtp3', 'lstm_wt',
                                   'conv_wd4', 'conv_wd', 'd_lstm_w', 'd_lstm_b', 'd_ln_w', 'd_ln_b', 'd_raw_conv')]


_tbound = None


def _tlib():
    global _tbound
    lib = L.get_lib()
    if _tbound is not lib:
        lib.rvt_stage_seq_train_fwd.restype = _i
        lib.rvt_stage_seq_train_fwd.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _vp]
        lib.rvt_stage_seq_bwd_ws_bytes.restype = ctypes.c_size_t
        lib.rvt_stage_seq_bwd_ws_bytes.argtypes = [_vp, _vp, _i, _i]
        lib.rvt_stage_seq_bwd.restype = _i
        lib.rvt_stage_seq_bwd.argtypes = [_vp] * 10 + [ctypes.c_size_t, _i, _i, _vp]
        _tbound = lib
    return lib


class TrainCall:
    """Everything one stage's training forward built for its backward: descriptors, host-side block arrays, the routes taken."""
    __slots__ = ('call', 'tr', 'saved_arr', 'tb_arr', 'inp', 'keep')


def train_routes(sw, g, dt, T: int, B: int, token_mask) -> Optional[dict]:
    """The routes stage_seq_forward / stage_seq_backward would take, or None where the C driver does not cover them."""
    from . import ops, tuning
    from .stage import use_attn_block, use_fused_mlp, use_lstm_scan, use_lstm_scan3
    if token_mask is not None or sw.dws is not None or tuning.get('route_wgrad_stream') != 0:
        return None
    C = g.C
    r = dict(attn_block=int(use_attn_block(dt, C, g.dim_head, g.ph * g.pw, training=True)))
    r['ln_linear'] = int(not r['attn_block'] and ops.ln_linear_supported(dt, C, 3 * C))
    if use_fused_mlp(dt, C, 'bwd_fused'):
        r['mlp_route'] = 1
    elif use_fused_mlp(dt, C, 'fwd_train'):
        return None                          # LDS-staged fused MLP that saves GELU / GELU': host loop
    else:
        r['mlp_route'] = 0
    r['mlp_bwd_both'] = int(r['mlp_route'] == 1 and ops.mlp_bwd_both_supported(dt, C))
    r['dgrad_ln_qkv'] = int(ops.linear_dgrad_ln_supported(dt, C, 3 * C))
    r['dgrad_ln_fc1'] = int(ops.linear_dgrad_ln_supported(dt, C, 4 * C))
    Ms = B * g.H * g.W
    if use_lstm_scan3(dt, C, sw.dws, T, True, Ms):
        r['lstm_route'] = 3
    elif use_lstm_scan(dt, C, sw.dws, T, True):
        r['lstm_route'] = 2 if ops.lstm_scan_saves_gates(dt, C) else 1
    else:
        r['lstm_route'] = 0
    r['lstm_scan_wgrad'] = int(r['lstm_route'] == 1 and tuning.get('route_lstm_scan_wgrad') != 0 and ops.lstm_scan_wgrad_supported(dt, C, Ms))
    r['conv_dgrad4'] = int(sw.conv_wd4 is not None and tuning.get('route_conv_dgrad4') != 0 and
                           ops.conv_dgrad4_supported(dt, g.H_in, g.W_in, g.Cin, C, g.k, g.stride, g.pad, T * B))
    # (the first block of a stage never has a norm1: maxvit_rnn.py:153 `skip_first_norm`; no token mask on this route)
    r['attn_preln'] = int((r['attn_block'] or ops.linear_dgrad_ln_supported(dt, C, 3 * C)) and tuning.get('route_attn_preln') != 0 and
                          sw.blocks[0][0]['n1_w'] is None)
    return r


def train_forward(sw, g, inp: torch.Tensor, h0, c0, T: int, B: int, routes: dict):
    """Training forward of one stage in ONE library call.  Returns (Hall, c_last, StageSaved) like stage.stage_seq_forward."""
    from . import ops
    from .stage import StageSaved
    dt, dev = sw.conv_w.dtype, inp.device
    H, W, C = g.H, g.W, g.C
    F_ = T * B
    E = lambda *shape, dtype=dt: torch.empty(shape, dtype=dtype, device=dev)
    u8 = inp.dtype == torch.uint8
    call = StageCall(sw, g, dt, u8, inp.shape[-2] if u8 else 0, inp.shape[-1] if u8 else 0)
    tr = RvtStageTrain()
    tr.struct_bytes = ctypes.sizeof(RvtStageTrain)
    for k, v in routes.items():
        setattr(tr, k, v)
    sv = StageSaved()
    y0, x0 = E(F_, H, W, C), E(F_, H, W, C)
    sv.inp, sv.y0, sv.mask = inp, y0, None
    nb = 2 * g.num_blocks
    saved_arr = (RvtBlockSaved * max(nb, 1))()
    x = x0
    i = 0
    for pair in sw.blocks:
        for bw in pair:
            has_n1 = bw['n1_w'] is not None
            s = dict(xin=x, qkv=None, a=E(F_, H, W, C), xmid=E(F_, H, W, C), hg=None, hgp=None, u=None, v2=None, hpre=False)
            if not routes['attn_block']:
                s['qkv'] = E(F_, H, W, 3 * C)
                s['u'] = E(F_, H, W, C) if has_n1 else x
            if routes['mlp_route'] == 0:
                s['v2'], s['hg'], s['hgp'] = E(F_, H, W, C), E(F_, H, W, 4 * C), E(F_, H, W, 4 * C)
            xout = E(F_, H, W, C)
            b = saved_arr[i]
            b.xin, b.a, b.xmid, b.xout = L.ptr(x), L.ptr(s['a']), L.ptr(s['xmid']), L.ptr(xout)
            b.qkv, b.u = L.ptr(s['qkv']), (L.ptr(s['u']) if (has_n1 and not routes['attn_block']) else None)
            b.v2, b.hg, b.hgp = L.ptr(s['v2']), L.ptr(s['hg']), L.ptr(s['hgp'])
            sv.blocks.append(s)
            x = xout
            i += 1
    Hall = E(T + 1, B, H, W, C)
    if h0 is None:
        Hall[0].zero_()
    else:
        Hall[0].copy_(h0)
    c_last = E(B, H, W, C, dtype=torch.float32)
    lr = routes['lstm_route']
    Csave = gates = Call = None
    if lr == 3:
        rows = ops.lstm_scan3_rows(C, B * H * W)
        Csave, gates = E(T, rows, C), E(T, rows, 4 * C)
        tr.lstm_wp3 = L.ptr(sw.scan3_packed(bwd=False))
    elif lr in (1, 2):
        Csave = E(T, B, H, W, C)
        gates = E(T, B, H, W, 4 * C) if lr == 2 else None
    else:
        Call = E(T + 1, B, H, W, C, dtype=torch.float32)
        if h0 is None:
            Call[0].zero_()
        else:
            Call[0].copy_(c0)
        gates = E(T, B, H, W, 4 * C)
    tr.saved = ctypes.cast(saved_arr, ctypes.POINTER(RvtBlockSaved))
    tr.y0, tr.x0, tr.Hall, tr.c_last = L.ptr(y0), L.ptr(x0), L.ptr(Hall), L.ptr(c_last)
    tr.Csave, tr.gates, tr.Call = L.ptr(Csave), L.ptr(gates), L.ptr(Call)
    lib = _tlib()
    rc = lib.rvt_stage_seq_train_fwd(ctypes.byref(call.desc), ctypes.byref(tr), L.ptr(inp), L.ptr(c0) if lr != 0 else None, T, B, L.stream_of(inp))
    if rc != 0:
        raise RuntimeError(f'rvt_stage_seq_train_fwd failed: {lib.rvt_last_error().decode()}')
    sv.x_last, sv.Hall, sv.Call, sv.gates = x, Hall, Call, gates
    sv.xin_lstm, sv.hconv, sv.Csave = x, None, Csave
    sv.c0 = None if (c0 is None or lr == 0) else c0.clone()      # (scan routes: the state the forward saw; RNNStates resets states in place)
    sv.scan3 = lr == 3
    tc = TrainCall()
    tc.call, tc.tr, tc.saved_arr, tc.tb_arr, tc.inp, tc.keep = call, tr, saved_arr, None, inp, (y0, x0)
    sv.train = tc
    if lr == 0:
        c_last = c_last      # (copied out of Call[T] by the library: BPTT keeps the T+1-slot array)
    return Hall, c_last, sv


def train_backward(sw, g, sv, dH, dc_last, T: int, B: int, need_input_grad: bool, prev_cot, sg, pre: str):
    """BPTT backward of one stage in ONE library call.  Returns (d_input or None, dh0, dc0) like stage.stage_seq_backward."""
    tc = sv.train
    tr, call = tc.tr, tc.call
    dt, dev = sv.y0.dtype, sv.y0.device
    H, W, C = g.H, g.W, g.C
    G = sg.g
    nb = 2 * g.num_blocks
    tb_arr = (RvtBlockTrain * max(nb, 1))()
    i = 0
    for pi, pair in enumerate(sw.blocks):
        for which, bw in enumerate(pair):
            bp = f'{pre}att_blocks.{pi}.{"att_window" if which == 0 else "att_grid"}.'
            b = tb_arr[i]
            b.qkv_wt, b.proj_wt, b.fc1_wt, b.fc2_wt = L.ptr(bw['qkv_wt']), L.ptr(bw['proj_wt']), L.ptr(bw['fc1_wt']), L.ptr(bw['fc2_wt'])
            has_n1 = bw['n1_w'] is not None
            b.d_n1_w, b.d_n1_b = (L.ptr(G(bp + 'norm1.weight')), L.ptr(G(bp + 'norm1.bias'))) if has_n1 else (None, None)
            b.d_qkv_w, b.d_qkv_b = L.ptr(G(bp + 'self_attn.qkv.weight')), L.ptr(G(bp + 'self_attn.qkv.bias'))
            b.d_S1, b.d_cs1 = L.ptr(G(bp + 'S1')), L.ptr(G(bp + 'cs1'))
            b.d_n2_w, b.d_n2_b = L.ptr(G(bp + 'norm2.weight')), L.ptr(G(bp + 'norm2.bias'))
            b.d_fc1_w, b.d_fc1_b = L.ptr(G(bp + 'mlp.net.0.0.weight')), L.ptr(G(bp + 'mlp.net.0.0.bias'))
            b.d_S2, b.d_cs2 = L.ptr(G(bp + 'S2')), L.ptr(G(bp + 'cs2'))
            i += 1
    tc.tb_arr = tb_arr
    tr.tb = ctypes.cast(tb_arr, ctypes.POINTER(RvtBlockTrain))
    tr.c0_saved = L.ptr(sv.c0)
    tr.lstm_wt, tr.conv_wd4, tr.conv_wd = L.ptr(sw.lstm_wt), L.ptr(sw.conv_wd4), L.ptr(getattr(sw, 'conv_wd', None))
    if tr.lstm_route == 3:
        tr.lstm_wtp3 = L.ptr(sw.scan3_packed(bwd=True))
    tr.d_lstm_w, tr.d_lstm_b = L.ptr(G(pre + 'lstm.conv1x1.weight')), L.ptr(G(pre + 'lstm.conv1x1.bias'))
    tr.d_ln_w, tr.d_ln_b = L.ptr(G(pre + 'downsample_cf2cl.norm.weight')), L.ptr(G(pre + 'downsample_cf2cl.norm.bias'))
    tr.d_raw_conv = L.ptr(G('raw/conv'))
    if dH is None and tr.lstm_route == 0:
        dH = torch.zeros((T, B, H, W, C), dtype=dt, device=dev)
    dcl = None if dc_last is None else dc_last.to(torch.float32).contiguous()
    dh0 = torch.empty((B, H, W, C), dtype=dt, device=dev)
    dc0 = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
    d_in = torch.empty((T * B, g.H_in, g.W_in, g.Cin), dtype=dt, device=dev) if need_input_grad else None
    lib = _tlib()
    st = L.stream_of(sv.y0)
    n = int(lib.rvt_stage_seq_bwd_ws_bytes(ctypes.byref(call.desc), ctypes.byref(tr), T, B))
    key = ('train', dev.type, dev.index, 0 if st is None else int(st))
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        _WS[key] = ws
    rc = lib.rvt_stage_seq_bwd(ctypes.byref(call.desc), ctypes.byref(tr), L.ptr(tc.inp), L.ptr(dH), L.ptr(dcl), L.ptr(prev_cot), L.ptr(d_in),
                               L.ptr(dh0), L.ptr(dc0), L.ptr(ws), ws.numel(), T, B, st)
    if rc != 0:
        raise RuntimeError(f'rvt_stage_seq_bwd failed: {lib.rvt_last_error().decode()}')
    return d_in, dh0, dc0
