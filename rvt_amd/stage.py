"""Stage-major execution of one backbone stage over a whole event-tensor sequence.

The reference runs time-major: for every time step, all four stages (modules/detection.py:131-148 →
maxvit_rnn.py:93-105).  Only the ConvLSTM is recurrent, and stage s at time t depends on stage s-1 at
the SAME t only, so here each stage processes all T·B frames at once (down-sampling conv + LayerNorm +
window block + grid block as large batched kernels) and then scans its ConvLSTM over t.  The result is
bit-for-bit the same dataflow; it only reorders independent work.  Backward runs stages 4→1, each as a
reverse ConvLSTM scan (BPTT) followed by the batched block / conv backward, so a stage's parameter
gradients are final as soon as that stage is done (used by rvt_amd.dist to overlap the all-reduce).

Layout: activations [T*B][H][W][C] channels-last in the compute dtype; LSTM cell state fp32.
`Hall`/`Call` carry T+1 time slots: slot 0 is the incoming state, slot t+1 the state after step t.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

import os

from . import ops
from .weights import StageWeights, unpack_conv_wgrad


def use_fused_mlp(dtype, C: int, what: str) -> bool:
    """Which MLP halves go through the fused kernels of csrc/mlp.hpp (built for C in {64,128}).
    Default = where they measured faster than the op-by-op chain on MI355X (profiles/microbench_mlp.py, bf16, ms, fused
    vs chain):
      C=64 : training forward 3.00 / 3.72, inference forward 2.03 / 3.72, backward dgrad chain 2.31 / 3.51 -> fused
      C=128: training forward 1.75 / 2.11, inference forward 1.53 / 2.11                                    -> fused
             backward dgrad chain 2.19 / 1.89 (one workgroup per CU: registers)                            -> chain
    The two directions are independent: the fused forward saves exactly what the chain backward reads (g, GELU', LN2 out).
    RVT_FUSED_MLP=1 forces every supported case (used by the parity tests), =0 disables all."""
    mode = os.environ.get('RVT_FUSED_MLP', 'auto')
    if mode == '0' or not ops.mlp_fused_supported(dtype, C):
        return False
    if mode == '1':
        return True
    return C == 64 or (C == 128 and what.startswith('fwd'))


class SideStream:
    """Weight-gradient GEMMs are off the critical path of backward (nothing downstream reads dW until the optimizer) and
    are read-only streams, while the input-gradient chain they hang off is write-heavy; running them on a second HIP
    stream lets the two share the chip.  Ordering is by events in both directions: run() makes the side stream wait for
    everything the main stream has issued so far (the producers of the operands), join() makes the main stream wait for
    the side stream.  Operands are kept alive here until join() instead of being record_stream()-ed: a recorded block
    sits in the caching allocator's event limbo after its last reference dies and the pool grows by gigabytes per stage
    (measured: 280 GiB reserved for 75 GiB live, then allocator retries of seconds inside a step)."""
    _streams = {}

    def __init__(self, like: Tensor):
        self.enabled = like.is_cuda and os.environ.get('RVT_WGRAD_STREAM', '1') == '1'
        self._keep = []
        if self.enabled:
            key = like.device.index
            if key not in SideStream._streams:
                SideStream._streams[key] = torch.cuda.Stream(device=like.device)
            self.stream = SideStream._streams[key]
            self.main = torch.cuda.current_stream(like.device)

    def run(self, fn, *operands):
        if not self.enabled:
            return fn()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self._keep.extend(t for t in operands if t is not None)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            return fn()

    def join(self):
        if self.enabled:
            self.main.wait_stream(self.stream)
            self._keep.clear()


@dataclass
class StageGeom:
    C: int
    Cin: int
    H_in: int
    W_in: int
    k: int
    stride: int
    pad: int
    ph: int
    pw: int
    dim_head: int
    num_blocks: int
    eps: float

    @property
    def H(self) -> int:
        return (self.H_in + 2 * self.pad - self.k) // self.stride + 1

    @property
    def W(self) -> int:
        return (self.W_in + 2 * self.pad - self.k) // self.stride + 1


class StageSaved:
    """Activations kept for backward (everything else is recomputed from these)."""
    __slots__ = ('inp', 'y0', 'blocks', 'x_last', 'Hall', 'Call', 'gates', 'mask', 'xin_lstm', 'hconv')

    def __init__(self):
        self.blocks: List[Dict[str, Tensor]] = []


def stage_seq_forward(sw: StageWeights, g: StageGeom, inp: Tensor, h0: Optional[Tensor], c0: Optional[Tensor],
                      T: int, B: int, save: bool, token_mask: Optional[Tensor] = None,
                      mask_token: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Optional[StageSaved]]:
    """inp: (T*B, H_in, W_in, cin_pad).  Returns Hall (T+1,B,H,W,C), Call (T+1,B,H,W,C) fp32, saved."""
    F_ = T * B
    H, W, C = g.H, g.W, g.C
    dt, dev = inp.dtype, inp.device
    sv = StageSaved() if save else None

    y0 = ops.conv_fwd(inp, sw.conv_w, g.k, g.stride, g.pad)                      # maxvit.py:175
    x = ops.layernorm_fwd(y0, sw.ln_w, sw.ln_b, g.eps)                            # maxvit.py:177
    mask_u8 = None
    if token_mask is not None:                                                    # maxvit_rnn.py:174-176
        mask_u8 = token_mask.reshape(F_ * H * W).to(torch.uint8).contiguous()
        ops.token_mask_fwd(x, mask_u8, mask_token.reshape(C).to(torch.float32).contiguous())
    if save:
        sv.inp, sv.y0, sv.mask = inp, y0, mask_u8

    for pair in sw.blocks:
        for bw, window in ((pair[0], True), (pair[1], False)):
            u = x if bw['n1_w'] is None else ops.layernorm_fwd(x, bw['n1_w'], bw['n1_b'], g.eps)
            qkv = ops.linear_fwd(u, bw['qkv_w'], bw['qkv_b'])                     # maxvit.py:347
            a = ops.attn_fwd(qkv, F_, H, W, C, g.dim_head, g.ph, g.pw, window)    # maxvit.py:349-352
            xmid = ops.linear_scale_res_fwd(a, bw['proj_w'], bw['proj_b'], bw['g1'], x)        # :353, :268
            v2 = None
            if use_fused_mlp(dt, C, 'fwd_train' if save else 'fwd_infer'):
                r = ops.mlp_fwd(xmid, bw['n2_w'], bw['n2_b'], bw['fc1_w'], bw['fc1_b'], bw['fc2_w'], bw['fc2_b'], bw['g2'],
                                g.eps, want_grad=save, want_v2=save)
                xout, hg, hgp = r[:3]
                v2 = r[3] if save else None     # LN2(xmid), saved for the fc1 weight gradient
            else:
                v2 = ops.layernorm_fwd(xmid, bw['n2_w'], bw['n2_b'], g.eps)
                # MLP fc1 + exact GELU; GELU' is saved too so backward never re-evaluates erf (maxvit.py:100-112)
                hg, hgp = ops.linear_gelu_fwd(v2, bw['fc1_w'], bw['fc1_b'], want_grad=save)
                xout = ops.linear_scale_res_fwd(hg, bw['fc2_w'], bw['fc2_b'], bw['g2'], xmid)            # :269
            if save:
                # the LayerNorm outputs are the B operands of the qkv / fc1 weight gradients: kept (1 row of C per token
                # each, 288 GB of HBM) rather than recomputed — a recompute is a read + a write + the re-read
                sv.blocks.append(dict(xin=x, qkv=qkv, a=a, xmid=xmid, hg=hg, hgp=hgp, u=u, v2=v2))
            x = xout

    Hall = torch.empty((T + 1, B, H, W, C), dtype=dt, device=dev)
    Call = torch.empty((T + 1, B, H, W, C), dtype=torch.float32, device=dev)
    if h0 is None:
        Hall[0].zero_()                                                           # rnn.py:43-47
        Call[0].zero_()
    else:
        Hall[0].copy_(h0)
        Call[0].copy_(c0)
    gates = torch.empty((T, B, H, W, 4 * C), dtype=dt, device=dev) if save else None
    # DWS-ConvLSTM (rnn.py:50-54): depth-wise 3x3 on h_{t-1} only, or on cat(x, h_{t-1}) (the x half batched over T)
    dws = sw.dws
    x_lstm = x
    hconv = None
    if dws is not None:
        hconv = torch.empty((T if save else 1, B, H, W, C), dtype=dt, device=dev)
        if not dws['only_hidden']:
            x_lstm = ops.dwconv(x, dws['w'][:C].contiguous(), dws['b'][:C].contiguous(), dws['k'])
    xt = x_lstm.view(T, B, H, W, C)
    for t in range(T):                                                            # rnn.py:52-67, one launch per step
        h_in = Hall[t]
        if dws is not None:
            wh = dws['w'] if dws['only_hidden'] else dws['w'][C:].contiguous()
            bh = dws['b'] if dws['only_hidden'] else dws['b'][C:].contiguous()
            h_in = ops.dwconv(Hall[t], wh, bh, dws['k'], out=hconv[t if save else 0])
        ops.lstm_fwd(xt[t], h_in, Call[t], sw.lstm_w, sw.lstm_b, Hall[t + 1], Call[t + 1],
                     gates[t] if save else None)
    if save:
        sv.x_last, sv.Hall, sv.Call, sv.gates = x, Hall, Call, gates
        sv.xin_lstm, sv.hconv = x_lstm, hconv
    return Hall, Call, sv


def stage_seq_backward(sw: StageWeights, g: StageGeom, sv: StageSaved, dH: Optional[Tensor], dc_last: Optional[Tensor],
                       T: int, B: int, need_input_grad: bool, prev_cot: Optional[Tensor],
                       p: Dict[str, Tensor], pre: str, side: Optional[SideStream] = None,
                       join: bool = True) -> Tuple[Optional[Tensor], Tensor, Tensor, Dict[str, Tensor]]:
    """dH: (T,B,H,W,C) cotangent of Hall[1:] (None = zeros); dc_last: (B,H,W,C) fp32 cotangent of Call[T].
    prev_cot: cotangent already attached to this stage's INPUT frames (T*B,H_in,W_in,Cin) (added to the conv dgrad).
    Returns (d_input or None, dh0, dc0, {param name: fp32 grad})."""
    F_ = T * B
    H, W, C = g.H, g.W, g.C
    dt, dev = sv.y0.dtype, sv.y0.device
    f32 = torch.float32
    grads: Dict[str, Tensor] = {}
    zeros = lambda *shape: torch.zeros(shape, dtype=f32, device=dev)

    # ---- ConvLSTM BPTT ------------------------------------------------------------------------------
    if dH is None:
        dH = torch.zeros((T, B, H, W, C), dtype=dt, device=dev)
    dz = torch.empty((T, B, H, W, 4 * C), dtype=dt, device=dev)
    dx = torch.empty((T, B, H, W, C), dtype=dt, device=dev)
    dc_rec = zeros(B, H, W, C) if dc_last is None else dc_last.to(f32).contiguous().clone()
    dh_rec = None
    dh_buf = [torch.empty((B, H, W, C), dtype=dt, device=dev) for _ in range(2)]
    dws = sw.dws
    dhc = torch.empty((T, B, H, W, C), dtype=dt, device=dev) if dws is not None else None   # d(dwconv(h_{t-1}))
    if dws is not None:
        wh = dws['w'] if dws['only_hidden'] else dws['w'][C:].contiguous()
    for t in range(T - 1, -1, -1):
        ops.lstm_gates_bwd(dH[t], dh_rec, dc_rec, sv.gates[t], sv.Call[t + 1], sv.Call[t], dz[t])
        nxt = dh_buf[t & 1]
        if dws is None:
            ops.lstm_dgrad(dz[t], sw.lstm_wt, dx[t], nxt)
        else:
            ops.lstm_dgrad(dz[t], sw.lstm_wt, dx[t], dhc[t])
            ops.dwconv(dhc[t], wh, None, dws['k'], transpose=True, out=nxt)
        dh_rec = nxt
    if side is None:
        side = SideStream(dz)
    h_seg = sv.Hall[:T].reshape(F_, H, W, C) if dws is None else sv.hconv.view(F_, H, W, C)

    def lstm_wgrad_fn():
        dwl = zeros(4 * C, 2 * C)
        dbl = zeros(4 * C)
        ops.lstm_wgrad(dz.view(F_, H, W, 4 * C), sv.xin_lstm, h_seg, dwl, dbl)
        grads[pre + 'lstm.conv1x1.weight'] = dwl.reshape(4 * C, 2 * C, 1, 1)
        grads[pre + 'lstm.conv1x1.bias'] = dbl
    side.run(lstm_wgrad_fn, dz, sv.xin_lstm, h_seg)
    dh0, dc0 = dh_rec, dc_rec
    dx = dx.view(F_, H, W, C)
    if dws is not None:
        kk = dws['k']
        cg = dws['w'].shape[0]
        dwd, dbd = zeros(cg, kk * kk), zeros(cg)
        hprev = sv.Hall[:T].reshape(F_, H, W, C)
        if dws['only_hidden']:
            ops.dwconv_wgrad(hprev, dhc.view(F_, H, W, C), dwd, dbd, kk)
        else:
            dwx, dbx, dwh, dbh = zeros(C, kk * kk), zeros(C), zeros(C, kk * kk), zeros(C)
            ops.dwconv_wgrad(sv.x_last, dx, dwx, dbx, kk)                    # dx here = d(dwconv_x(x))
            ops.dwconv_wgrad(hprev, dhc.view(F_, H, W, C), dwh, dbh, kk)
            dwd, dbd = torch.cat([dwx, dwh]), torch.cat([dbx, dbh])
            dx = ops.dwconv(dx, dws['w'][:C].contiguous(), None, kk, transpose=True)
        grads[pre + 'lstm.conv3x3_dws.weight'] = dwd.reshape(cg, 1, kk, kk)
        grads[pre + 'lstm.conv3x3_dws.bias'] = dbd
    del dz

    # ---- attention blocks, reversed ---------------------------------------------------------------------
    bi_flat = len(sv.blocks)
    for pi in range(len(sw.blocks) - 1, -1, -1):
        for which in (1, 0):
            bw = sw.blocks[pi][which]
            window = which == 0
            bi_flat -= 1
            s = sv.blocks[bi_flat]
            bp = f'{pre}att_blocks.{pi}.{"att_window" if window else "att_grid"}.'
            # MLP branch: xout = xmid + g2 * (gelu(hd) W2^T + b2)
            def fc2_wgrad_fn(dx=dx, s=s, bw=bw, bp=bp):
                S2 = zeros(C, 4 * C)
                cs = zeros(C)
                ops.linear_wgrad(dx, s['hg'], S2, colsum_out=cs)
                grads[bp + 'mlp.net.2.weight'] = bw['g2'][:, None] * S2
                grads[bp + 'mlp.net.2.bias'] = bw['g2'] * cs
                grads[bp + 'ls2.gamma'] = (bw['fc2_w32'] * S2).sum(1) + p[bp + 'mlp.net.2.bias'].detach().to(f32) * cs
            side.run(fc2_wgrad_fn, dx, s['hg'])
            dn2w, dn2b = zeros(C), zeros(C)
            fused = use_fused_mlp(dt, C, 'bwd')
            if fused:       # fc2 dgrad * gp, fc1 dgrad and LayerNorm-2 backward (+ residual) in one kernel
                dhd, dxmid = ops.mlp_bwd_dgrad(dx, s['hgp'], s['xmid'], bw['n2_w'], bw['fc2_wt'], bw['fc1_wt'], dn2w, dn2b,
                                               g.eps)
            else:
                dhd = ops.linear_dgrad(dx, bw['fc2_wt'], mul=s['hgp'])
            def fc1_wgrad_fn(dhd=dhd, s=s, bw=bw, bp=bp):
                v2 = s['v2'] if s['v2'] is not None else ops.layernorm_fwd(s['xmid'], bw['n2_w'], bw['n2_b'], g.eps)
                dW1 = zeros(4 * C, C)
                db1 = zeros(4 * C)
                ops.linear_wgrad(dhd, v2, dW1, colsum_out=db1)
                grads[bp + 'mlp.net.0.0.weight'] = dW1
                grads[bp + 'mlp.net.0.0.bias'] = db1
            side.run(fc1_wgrad_fn, dhd, s['xmid'])
            if not fused:
                dv2 = ops.linear_dgrad(dhd, bw['fc1_wt'])
                dxmid = ops.layernorm_bwd(s['xmid'], bw['n2_w'], dv2, dx, dn2w, dn2b, g.eps)
                del dv2
            del dhd
            grads[bp + 'norm2.weight'] = dn2w
            grads[bp + 'norm2.bias'] = dn2b
            # attention branch: xmid = xin + g1 * (a Wp^T + bp)
            def proj_wgrad_fn(dxmid=dxmid, s=s, bw=bw, bp=bp):
                S1 = zeros(C, C)
                cs1 = zeros(C)
                ops.linear_wgrad(dxmid, s['a'], S1, colsum_out=cs1)
                grads[bp + 'self_attn.proj.weight'] = bw['g1'][:, None] * S1
                grads[bp + 'self_attn.proj.bias'] = bw['g1'] * cs1
                grads[bp + 'ls1.gamma'] = (bw['proj_w32'] * S1).sum(1) + p[bp + 'self_attn.proj.bias'].detach().to(f32) * cs1
            side.run(proj_wgrad_fn, dxmid, s['a'])
            da = ops.linear_dgrad(dxmid, bw['proj_wt'])
            dqkv = ops.attn_bwd(s['qkv'], da, F_, H, W, C, g.dim_head, g.ph, g.pw, window)
            del da
            def qkv_wgrad_fn(dqkv=dqkv, s=s, bw=bw, bp=bp):
                u = s['u']
                dWq = zeros(3 * C, C)
                dbq = zeros(3 * C)
                ops.linear_wgrad(dqkv, u, dWq, colsum_out=dbq)
                grads[bp + 'self_attn.qkv.weight'] = dWq
                grads[bp + 'self_attn.qkv.bias'] = dbq
            side.run(qkv_wgrad_fn, dqkv, s['xin'])
            if bw['n1_w'] is None:
                dx = ops.linear_dgrad(dqkv, bw['qkv_wt'], add=dxmid)
            else:
                du = ops.linear_dgrad(dqkv, bw['qkv_wt'])
                dn1w, dn1b = zeros(C), zeros(C)
                dx = ops.layernorm_bwd(s['xin'], bw['n1_w'], du, dxmid, dn1w, dn1b, g.eps)
                grads[bp + 'norm1.weight'] = dn1w
                grads[bp + 'norm1.bias'] = dn1b
                del du
            del dqkv, dxmid

    # ---- token mask, down-sampling LayerNorm + conv ---------------------------------------------------------
    if sv.mask is not None:
        dtok = zeros(C)
        ops.token_mask_bwd(dx, sv.mask, dtok)                                     # also zeroes dx on masked tokens
        grads[pre + 'mask_token'] = dtok.reshape(1, 1, 1, C)
    dlw, dlb = zeros(C), zeros(C)
    dy0 = ops.layernorm_bwd(sv.y0, sw.ln_w, dx, None, dlw, dlb, g.eps)
    grads[pre + 'downsample_cf2cl.norm.weight'] = dlw
    grads[pre + 'downsample_cf2cl.norm.bias'] = dlb
    def conv_wgrad_fn():
        dwc = zeros(C, g.k * g.k * sw.cin_pad)
        ops.conv_wgrad(sv.inp, dy0, dwc, g.k, g.stride, g.pad)
        grads[pre + 'downsample_cf2cl.conv.weight'] = unpack_conv_wgrad(dwc, g.Cin, g.k)
    side.run(conv_wgrad_fn, sv.inp, dy0)
    d_in = None
    if need_input_grad:
        d_in = ops.conv_dgrad(dy0, sw.conv_wd, prev_cot, g.H_in, g.W_in, g.Cin, g.k, g.stride, g.pad)
    if join:
        side.join()             # every parameter gradient of this stage is final from here on (DDP hook, optimizer)
    return d_in, dh0, dc0, grads
