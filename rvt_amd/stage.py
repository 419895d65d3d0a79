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

from . import ops, tuning
from .weights import StageGrads, StageWeights



def use_fused_mlp(dtype, C: int, what: str) -> bool:
    """Which MLP halves go through the fused kernels of csrc/mlp.hpp / mlp_chain.hpp.
      'bwd_fused' (C = 64): the whole backward — recompute of LN2 / fc1 / GELU, both input-gradient products, LayerNorm
          backward and the weight gradients (two launches) — from (dxout, xmid) alone; the forward then saves
          nothing but the block input (3 + 2 rows of C per token through HBM for the MLP half instead of 32).
      'fwd_train' / 'fwd_infer' / 'bwd' (C in {64,128}): fused forward (optionally saving GELU, GELU', LN2 out) and the
          fused input-gradient chain.  Measured on MI355X (profiles/microbench_mlp.py, bf16, ms, fused vs op-by-op chain):
          C=64 : training forward 3.00 / 3.72, inference forward 2.03 / 3.72, backward dgrad chain 2.31 / 3.51 -> fused
          C=128: training forward 1.75 / 2.11, inference forward 1.53 / 2.11                                    -> fused
                 backward dgrad chain 2.19 / 1.89 (one workgroup per CU: registers)                            -> chain
    tuning.route_fused_mlp = 1 forces every supported case (used by the parity tests), 0 disables all;
    tuning.route_mlp_bwd_fused = 0 disables only the everything-on-chip backward."""
    mode = tuning.get('route_fused_mlp')
    if what == 'bwd_fused':
        return mode != 0 and tuning.get('route_mlp_bwd_fused') != 0 and ops.mlp_bwd_fused_supported(dtype, C)
    if mode == 0 or not ops.mlp_fused_supported(dtype, C):
        return False
    if mode == 1:
        return True
    return C == 64 or (C == 128 and what.startswith('fwd'))


def use_attn_block(dtype, C: int, dh: int, n_tok: int, training: bool = False) -> bool:
    """Attention half of a block (norm1, qkv, partition attention, proj, LayerScale + residual) as ONE kernel per direction
    (csrc/attn_block.hpp, one wave per partition) instead of LayerNorm + linear + attention core + linear (+ their
    backward chain): where it is built (C = 64, dim_head 32, partitions of 33..96 tokens).  tuning.route_attn_block = 0 disables."""
    # partitions of more than 64 tokens (Gen1: 8 x 10) have a fused forward only; a forward that keeps activations for a
    # backward therefore takes the op-by-op chain there
    if training and n_tok > 64:
        return False
    return tuning.get('route_attn_block') != 0 and ops.attn_block_supported(dtype, C, dh, n_tok)


def use_lstm_scan(dtype, C: int, dws, T: int = 0, save: bool = True) -> bool:
    """ConvLSTM with the time loop inside the kernel (csrc/lstm_scan.hpp) instead of one launch per step: only the 1x1-conv
    cell (dws_conv False — every shipped config); by default where the weights stay resident in LDS (C <= 64), C = 128
    (weights streamed from L2) with tuning.route_lstm_scan = 1 (all supported widths; the parity tests); 0 disables."""
    mode = tuning.get('route_lstm_scan')
    if mode == 0 or dws is not None or not ops.lstm_scan_supported(dtype, C):
        return False
    if mode == -1 and T == 1 and not save:
        return False                     # one no-grad step (streaming inference): the per-step GEMM beats staging the scan's weights
    return True if mode == 1 else (C <= 64 or ops.lstm_scan_saves_gates(dtype, C))


def use_lstm_scan3(dtype, C: int, dws, T: int = 0, save: bool = True, M: int = 1 << 30) -> bool:
    """ConvLSTM (bf16, C = 128 / 256) with the time loop in the kernel, the weights streamed from L2 in operand order and the gates
    saved for the reverse scan (csrc/lstm_scan3.hpp): instead of 3 launches per step at C = 256 (weights too large for the chip),
    instead of the register-resident-weight scan of lstm_scan.hpp at C = 128.
    One no-grad step (streaming inference, T = 1) keeps the per-step GEMM: packing + streaming the weights buys nothing there."""
    if dws is not None or not ops.lstm_scan3_supported(dtype, C):
        return False
    if C == 128 and M < 16384 and tuning.get('route_lstm_scan') != 0:
        # few tokens per step (stage 3 of RVT-Tiny: 2560): the register-resident weights of lstm_scan.hpp win - 0.118 + 0.172 ms
        # against 0.124 + 0.198; at 92160 tokens (stage 2 of RVT-Base) the streamed form does: 1.28 + 1.66 against 1.50 + 2.15
        return False
    return save or T > 1


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
        # off by default since round 2: with the fused stage-1 kernels the step on ONE stream costs exactly the sum of its
        # kernels' isolated times (98.6 ms, profiles/r2/op_breakdown_serial_r2i.txt) and the second stream only adds
        # contention (100.7 ms); tuning.route_wgrad_stream = 1 re-enables it.
        # route_wgrad_stream = 2 (round 5): DEFERRED - the weight-gradient launches of a stage whose successor in the backward
        # order opens with a per-step reverse ConvLSTM scan (42 launches of 23 - 90 tiles on 256 CUs) are queued and start on
        # the side stream when that scan starts, i.e. they fill a chip that is two thirds idle instead of competing with full grids.
        mode = tuning.get('route_wgrad_stream') if like.is_cuda else 0
        self.enabled = mode in (1, 2)
        self.defer_mode = mode == 2
        self.deferring = False          # set per stage by the backward driver (rvt_amd/backbone.py)
        self._queue = []
        self._flushed = False
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
        if self.defer_mode:
            if not self.deferring:
                return fn()
            self._queue.append(fn)
            self._keep.extend(t for t in operands if t is not None)
            return None
        ev = torch.cuda.Event()
        ev.record(self.main)
        self._keep.extend(t for t in operands if t is not None)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            return fn()

    def flush(self) -> bool:
        """Deferred mode: start the queued launches on the side stream, ordered after everything the main stream has issued."""
        if not (self.enabled and self.defer_mode and self._queue):
            return False
        ev = torch.cuda.Event()
        ev.record(self.main)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            for fn in self._queue:
                fn()
        # the closures hold the last references to what their launches read (the stage's saved activations: backbone.py has
        # already dropped ctx.svs[si]); they stay alive until join() has ordered the main stream behind the side stream -
        # dropping them here would hand the blocks back to the caching allocator while the side-stream kernels are pending
        self._keep.extend(self._queue)
        self._queue = []
        self._flushed = True
        return True

    def join(self):
        if not self.enabled:
            return
        if self.defer_mode:
            if self._flushed:
                self.main.wait_stream(self.stream)
                self._flushed = False
                if not self._queue:
                    self._keep.clear()
            return
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
    __slots__ = ('inp', 'y0', 'blocks', 'x_last', 'Hall', 'Call', 'gates', 'mask', 'xin_lstm', 'hconv', 'Csave', 'c0', 'scan3', 'train', '__weakref__')

    def __init__(self):
        self.blocks: List[Dict[str, Tensor]] = []
        self.scan3 = False
        self.train = None           # set by the C-side training driver (rvt_amd/stage_driver.py: train_forward)


def stage_seq_forward(sw: StageWeights, g: StageGeom, inp: Tensor, h0: Optional[Tensor], c0: Optional[Tensor],
                      T: int, B: int, save: bool, token_mask: Optional[Tensor] = None,
                      mask_token: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Optional[StageSaved]]:
    """inp: (T*B, H_in, W_in, cin_pad), or the uint8 planes (T*B, Cin, h, w) where ops.stem_supported.  Returns Hall (T+1,B,H,W,C), the final cell state (B,H,W,C) fp32, saved."""
    F_ = T * B
    H, W, C = g.H, g.W, g.C
    dt, dev = sw.conv_w.dtype, inp.device
    if save and tuning.get('route_stage_driver_train') != 0:
        # the whole training forward of the stage as ONE library call (include/rvt_hip.h: rvt_stage_seq_train_fwd) where covered
        from . import stage_driver
        routes = stage_driver.train_routes(sw, g, dt, T, B, token_mask)
        if routes is not None:
            return stage_driver.train_forward(sw, g, inp, h0, c0, T, B, routes)
    sv = StageSaved() if save else None

    if inp.dtype == torch.uint8:
        # first stage on the loader's planes (T*B, Cin, h, w): cast + pad + conv + LayerNorm in one launch (csrc/stem.hpp)
        y0, x = ops.stem_fwd(inp, sw.conv_w, sw.ln_w, sw.ln_b, g.H_in, g.W_in, g.eps)
    else:
        y0 = ops.conv_fwd(inp, sw.conv_w, g.k, g.stride, g.pad)                  # maxvit.py:175
        x = ops.layernorm_fwd(y0, sw.ln_w, sw.ln_b, g.eps)                        # maxvit.py:177
    mask_u8 = None
    if token_mask is not None:                                                    # maxvit_rnn.py:174-176
        mask_u8 = token_mask.reshape(F_ * H * W).to(torch.uint8).contiguous()
        ops.token_mask_fwd(x, mask_u8, mask_token.reshape(C).to(torch.float32).contiguous())
    if save:
        sv.inp, sv.y0, sv.mask = inp, y0, mask_u8

    for pair in sw.blocks:
        for bw, window in ((pair[0], True), (pair[1], False)):
            if use_attn_block(dt, C, g.dim_head, g.ph * g.pw, training=save):
                # maxvit.py:268 in one launch; backward recomputes q / k / v / P from the block input (nothing but `a`,
                # the operand of the proj weight gradient, is kept)
                xmid, a = ops.attn_block_fwd(x, bw['n1_w'], bw['n1_b'], bw['qkv_w'], bw['qkv_b'], bw['proj_w'], bw['proj_b'],
                                             bw['g1'], F_, H, W, C, g.dim_head, g.ph, g.pw, window, g.eps, want_a=save)
                u = qkv = None
            else:
                if ops.ln_linear_supported(dt, C, 3 * C):
                    # norm1 + qkv in one launch (csrc/ln_linear.hpp; C = 128); u is kept for the qkv weight gradient
                    u, qkv = ops.ln_linear_fwd(x, bw['n1_w'], bw['n1_b'], bw['qkv_w'], bw['qkv_b'], g.eps, want_u=save)
                    if bw['n1_w'] is None:
                        u = x
                else:
                    u = x if bw['n1_w'] is None else ops.layernorm_fwd(x, bw['n1_w'], bw['n1_b'], g.eps)
                    qkv = ops.linear_fwd(u, bw['qkv_w'], bw['qkv_b'])                 # maxvit.py:347
                a = ops.attn_fwd(qkv, F_, H, W, C, g.dim_head, g.ph, g.pw, window)    # maxvit.py:349-352
                xmid = ops.linear_scale_res_fwd(a, bw['proj_w'], bw['proj_b'], bw['g1'], x)        # :353, :268
            v2 = None
            hpre = False
            if use_fused_mlp(dt, C, 'bwd_fused'):
                # the backward recomputes everything from xmid: inference-flavoured forward, nothing else kept (the no-grad
                # forward takes the same kernel, so eval and training outputs are bit-identical)
                xout, hg, hgp = ops.mlp_fwd(xmid, bw['n2_w'], bw['n2_b'], bw['fc1_w'], bw['fc1_b'], bw['fc2_w'], bw['fc2_b'],
                                            bw['g2'], g.eps, want_grad=False)
            elif use_fused_mlp(dt, C, 'fwd_train' if save else 'fwd_infer'):
                # training forward at C = 128: by default only the pre-activation h is kept (hg = h, hgp = None); the backward
                # applies GELU on load in the fc2 weight gradient and GELU' in the epilogue of the fc2 input gradient
                pre = save and tuning.get('route_mlp_store_pre') != 0 and not use_fused_mlp(dt, C, 'bwd')
                r = ops.mlp_fwd(xmid, bw['n2_w'], bw['n2_b'], bw['fc1_w'], bw['fc1_b'], bw['fc2_w'], bw['fc2_b'], bw['g2'],
                                g.eps, want_grad=save and not pre, want_v2=save, want_pre=pre)
                xout, hg, hgp = r[:3]
                hpre = pre
                v2 = r[3] if save else None     # LN2(xmid), saved for the fc1 weight gradient
            else:
                v2 = ops.layernorm_fwd(xmid, bw['n2_w'], bw['n2_b'], g.eps)
                # MLP fc1 + exact GELU; GELU' is saved too so backward never re-evaluates erf (maxvit.py:100-112)
                hg, hgp = ops.linear_gelu_fwd(v2, bw['fc1_w'], bw['fc1_b'], want_grad=save)
                xout = ops.linear_scale_res_fwd(hg, bw['fc2_w'], bw['fc2_b'], bw['g2'], xmid)            # :269
            if save:
                # the LayerNorm outputs are the B operands of the qkv / fc1 weight gradients: kept (1 row of C per token
                # each, 288 GB of HBM) rather than recomputed — a recompute is a read + a write + the re-read
                sv.blocks.append(dict(xin=x, qkv=qkv, a=a, xmid=xmid, hg=hg, hgp=hgp, u=u, v2=v2, hpre=hpre))
            x = xout

    Hall = torch.empty((T + 1, B, H, W, C), dtype=dt, device=dev)
    dws = sw.dws
    # a no-grad forward on the per-step route reads the incoming states where they are (streaming inference, T = 1: the two
    # state copies per stage were 5 % of the step); BPTT and the scan kernel want them in slot 0
    scan3 = use_lstm_scan3(dt, C, dws, T, save, B * H * W)
    direct0 = (not save) and h0 is not None and not scan3 and not use_lstm_scan(dt, C, dws, T, save) and h0.dtype == dt and h0.is_contiguous() \
        and c0 is not None and c0.dtype == torch.float32 and c0.is_contiguous()
    if h0 is None:
        Hall[0].zero_()                                                           # rnn.py:43-47
    elif not direct0:
        Hall[0].copy_(h0)
    if scan3:
        # wide stage: all T steps in ONE launch, weights streamed in operand order, gates + cell states saved in dump order
        c_last = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
        rows = ops.lstm_scan3_rows(C, B * H * W)
        Csave = torch.empty((T, rows, C), dtype=dt, device=dev) if save else None
        gsave = torch.empty((T, rows, 4 * C), dtype=dt, device=dev) if save else None
        ops.lstm_scan3_fwd(x.view(T, B, H, W, C), Hall, c0, c_last, Csave, sw.scan3_packed(bwd=False), sw.lstm_bn, gsave)
        if save:
            sv.x_last, sv.Hall, sv.Call, sv.gates = x, Hall, None, gsave
            sv.xin_lstm, sv.hconv, sv.Csave, sv.c0 = x, None, Csave, (None if c0 is None else c0.clone())
            sv.scan3 = True
        return Hall, c_last, sv
    if use_lstm_scan(dt, C, dws, T, save):
        # all T steps in ONE launch: h / c stay on chip, BPTT keeps only a T-typed copy of the cell states
        c_last = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
        Csave = torch.empty((T, B, H, W, C), dtype=dt, device=dev) if save else None
        # C = 128 (bf16): weights in the register file; the gates are stored for a reverse scan that keeps W^T in registers
        gsave = torch.empty((T, B, H, W, 4 * C), dtype=dt, device=dev) if save and ops.lstm_scan_saves_gates(dt, C) else None
        ops.lstm_scan_fwd(x.view(T, B, H, W, C), Hall, c0, c_last, Csave, sw.lstm_wn, sw.lstm_bn, gates_out=gsave)
        if save:
            sv.x_last, sv.Hall, sv.Call, sv.gates = x, Hall, None, gsave
            # the incoming cell state may alias the caller's tensor (RNNStates resets states in place, modules/utils/detection.py:96-113):
            # BPTT needs the value the forward saw, so it keeps a copy (B*H*W*C fp32 per stage)
            sv.xin_lstm, sv.hconv, sv.Csave, sv.c0 = x, None, Csave, (None if c0 is None else c0.clone())
        return Hall, c_last, sv
    # cell states: all T+1 slots are kept for BPTT; a no-grad forward ping-pongs between two
    nc = T + 1 if save else 2
    Call = torch.empty((nc, B, H, W, C), dtype=torch.float32, device=dev)
    if h0 is None:
        Call[0].zero_()
    elif not direct0:
        Call[0].copy_(c0)
    gates = torch.empty((T, B, H, W, 4 * C), dtype=dt, device=dev) if save else None
    # DWS-ConvLSTM (rnn.py:50-54): depth-wise 3x3 on h_{t-1} only, or on cat(x, h_{t-1}) (the x half batched over T)
    x_lstm = x
    hconv = None
    if dws is not None:
        hconv = torch.empty((T if save else 1, B, H, W, C), dtype=dt, device=dev)
        if not dws['only_hidden']:
            x_lstm = ops.dwconv(x, dws['w'][:C], dws['b'][:C], dws['k'])
    xt = x_lstm.view(T, B, H, W, C)
    for t in range(T):                                                            # rnn.py:52-67, one launch per step
        h_prev = h0.view(B, H, W, C) if (direct0 and t == 0) else Hall[t]
        c_prev = c0.view(B, H, W, C) if (direct0 and t == 0) else Call[t % nc]
        h_in = h_prev
        if dws is not None:
            wh = dws['w'] if dws['only_hidden'] else dws['w'][C:]
            bh = dws['b'] if dws['only_hidden'] else dws['b'][C:]
            h_in = ops.dwconv(h_prev, wh, bh, dws['k'], out=hconv[t if save else 0])
        ops.lstm_fwd(xt[t], h_in, c_prev, sw.lstm_w, sw.lstm_b, Hall[t + 1], Call[(t + 1) % nc],
                     gates[t] if save else None)
    # the final cell state is handed to the caller (RNNStates keeps it across steps): a buffer of its own when the
    # T+1-slot array is what BPTT holds on to
    c_last = Call[T % nc].clone() if save else Call[T % nc]
    if save:
        sv.x_last, sv.Hall, sv.Call, sv.gates = x, Hall, Call, gates
        sv.xin_lstm, sv.hconv, sv.Csave, sv.c0 = x_lstm, hconv, None, None
    return Hall, c_last, sv


def stage_seq_backward(sw: StageWeights, g: StageGeom, sv: StageSaved, dH: Optional[Tensor], dc_last: Optional[Tensor],
                       T: int, B: int, need_input_grad: bool, prev_cot: Optional[Tensor],
                       sg: StageGrads, pre: str, side: Optional[SideStream] = None,
                       finalize=None) -> Tuple[Optional[Tensor], Tensor, Tensor]:
    """dH: (T,B,H,W,C) cotangent of Hall[1:] (None = zeros); dc_last: (B,H,W,C) fp32 cotangent of Call[T].
    prev_cot: cotangent already attached to this stage's INPUT frames (T*B,H_in,W_in,Cin) (added to the conv dgrad).
    Parameter gradients are ACCUMULATED into the views of the stage's fp32 bucket `sg` (rvt_amd/weights.py); `finalize`
    (LayerScale fold + conv unpack, two table launches) runs behind the weight-gradient GEMMs on their stream.
    Returns (d_input or None, dh0, dc0)."""
    F_ = T * B
    H, W, C = g.H, g.W, g.C
    dt, dev = sv.y0.dtype, sv.y0.device
    f32 = torch.float32
    G = sg.g
    if sv.train is not None:
        # BPTT + block / conv backward of the stage as ONE library call (rvt_stage_seq_bwd); the LayerScale fold / conv unpack follow
        from . import stage_driver
        out = stage_driver.train_backward(sw, g, sv, dH, dc_last, T, B, need_input_grad, prev_cot, sg, pre)
        if finalize is not None:
            finalize()
        return out

    # ---- ConvLSTM BPTT ------------------------------------------------------------------------------
    dx = torch.empty((T, B, H, W, C), dtype=dt, device=dev)
    dws = sw.dws
    lstm_wgrad_done = False
    dz = None
    if sv.scan3:
        # reverse scan of a wide stage in ONE launch on the saved gates; dz goes to the weight-gradient GEMM below
        dh_rec = torch.empty((B, H, W, C), dtype=dt, device=dev)
        dc_rec = torch.empty((B, H, W, C), dtype=f32, device=dev)
        dz = torch.empty((T, B, H, W, 4 * C), dtype=dt, device=dev)
        ops.lstm_scan3_bwd(sv.gates, sv.Csave, sv.c0, dH, None if dc_last is None else dc_last.to(f32).contiguous(),
                           sw.scan3_packed(bwd=True), dx, dz, dh_rec, dc_rec)
    elif sv.Csave is not None:
        # reverse scan in ONE launch: gates recomputed from (x_t, h_{t-1}), dc / dh_rec in registers across t; where built
        # (bf16, C <= 64) the weight gradients are accumulated in the same kernel and dz never exists in HBM
        dh_rec = torch.empty((B, H, W, C), dtype=dt, device=dev)
        dc_rec = torch.empty((B, H, W, C), dtype=f32, device=dev)
        lstm_wgrad_done = sv.gates is None and tuning.get('route_lstm_scan_wgrad') != 0 and \
            ops.lstm_scan_wgrad_supported(dt, C, B * H * W)
        if not lstm_wgrad_done:
            dz = torch.empty((T, B, H, W, 4 * C), dtype=dt, device=dev)
        ops.lstm_scan_bwd(sv.xin_lstm.view(T, B, H, W, C), sv.Hall, sv.Csave, sv.c0, dH,
                          None if dc_last is None else dc_last.to(f32).contiguous(), sw.lstm_wn, sw.lstm_wt, sw.lstm_bn,
                          dx, dz, dh_rec, dc_rec,
                          dw=G(pre + 'lstm.conv1x1.weight').view(4 * C, 2 * C) if lstm_wgrad_done else None,
                          db=G(pre + 'lstm.conv1x1.bias') if lstm_wgrad_done else None, gates=sv.gates)
    else:
        dz = torch.empty((T, B, H, W, 4 * C), dtype=dt, device=dev)
        if dH is None:
            dH = torch.zeros((T, B, H, W, C), dtype=dt, device=dev)
        dc_rec = torch.zeros((B, H, W, C), dtype=f32, device=dev) if dc_last is None else dc_last.to(f32).contiguous().clone()
        dh_rec = None
        dh_buf = [torch.empty((B, H, W, C), dtype=dt, device=dev) for _ in range(2)]
        dhc = torch.empty((T, B, H, W, C), dtype=dt, device=dev) if dws is not None else None   # d(dwconv(h_{t-1}))
        if dws is not None:
            wh = dws['w'] if dws['only_hidden'] else dws['w'][C:]
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
        side = SideStream(dx)
    h_seg = sv.Hall[:T].reshape(F_, H, W, C) if dws is None else sv.hconv.view(F_, H, W, C)

    def lstm_wgrad_fn(dz=dz, h_seg=h_seg):          # (bound now: `dz` is deleted below and the launch may be deferred)
        ops.lstm_wgrad(dz.view(F_, H, W, 4 * C), sv.xin_lstm, h_seg, G(pre + 'lstm.conv1x1.weight').view(4 * C, 2 * C),
                       G(pre + 'lstm.conv1x1.bias'))
    if not lstm_wgrad_done:
        side.run(lstm_wgrad_fn, dz, sv.xin_lstm, h_seg)
    dh0, dc0 = dh_rec, dc_rec
    dx = dx.view(F_, H, W, C)
    if dws is not None:
        kk = dws['k']
        cg = dws['w'].shape[0]
        dwd, dbd = G(pre + 'lstm.conv3x3_dws.weight').view(cg, kk * kk), G(pre + 'lstm.conv3x3_dws.bias')
        hprev = sv.Hall[:T].reshape(F_, H, W, C)
        if dws['only_hidden']:
            ops.dwconv_wgrad(hprev, dhc.view(F_, H, W, C), dwd, dbd, kk)
        else:
            ops.dwconv_wgrad(sv.x_last, dx, dwd[:C], dbd[:C], kk)                 # dx here = d(dwconv_x(x))
            ops.dwconv_wgrad(hprev, dhc.view(F_, H, W, C), dwd[C:], dbd[C:], kk)
            dx = ops.dwconv(dx, dws['w'][:C], None, kk, transpose=True)
    del dz

    # ---- attention blocks, reversed ---------------------------------------------------------------------
    dy0_fused = None
    bi_flat = len(sv.blocks)
    for pi in range(len(sw.blocks) - 1, -1, -1):
        for which in (1, 0):
            bw = sw.blocks[pi][which]
            window = which == 0
            bi_flat -= 1
            s = sv.blocks[bi_flat]
            bp = f'{pre}att_blocks.{pi}.{"att_window" if window else "att_grid"}.'
            # MLP branch: xout = xmid + g2 * (gelu(hd) W2^T + b2).  The weight-gradient GEMM delivers the raw products
            # S2 = dxout^T g and cs2 = colsum(dxout); LayerScale is folded in by the finalize table launch.
            dn2w, dn2b = G(bp + 'norm2.weight'), G(bp + 'norm2.bias')
            if s['hg'] is None:
                # everything on chip: recompute, both input-gradient products, LayerNorm backward and the fc1 / fc2 weight
                # gradients (accumulated in registers) in one kernel; nothing for the weight-gradient stream to do
                # two launches, cut along the critical path: the input-gradient half here, the weight-gradient half (which
                # needs all 2 x 64 x 256 accumulators) on the weight-gradient stream
                # (round 4: ONE launch where the library supports it — the weight-gradient kernel hands dh to two waves that form
                # dh W1, LayerNorm backward in the staging role; otherwise)
                if ops.mlp_bwd_both_supported(dt, C):
                    dxmid = ops.mlp_bwd_recompute_both(dx, s['xmid'], bw['n2_w'], bw['n2_b'], bw['fc1_w'], bw['fc1_b'], bw['fc2_wt'],
                                                       bw['fc1_wt'], dn2w, dn2b, G(bp + 'mlp.net.0.0.weight'),
                                                       G(bp + 'mlp.net.0.0.bias'), G(bp + 'S2'), G(bp + 'cs2'), g.eps)
                else:
                    def mlp_wgrad_fn(dx=dx, s=s, bw=bw, bp=bp):
                        ops.mlp_bwd_recompute_wgrad(dx, s['xmid'], bw['n2_w'], bw['n2_b'], bw['fc1_w'], bw['fc1_b'],
                                                    bw['fc2_wt'], G(bp + 'mlp.net.0.0.weight'), G(bp + 'mlp.net.0.0.bias'),
                                                    G(bp + 'S2'), G(bp + 'cs2'), g.eps)
                    side.run(mlp_wgrad_fn, dx, s['xmid'])
                    dxmid = ops.mlp_bwd_recompute_dgrad(dx, s['xmid'], bw['n2_w'], bw['n2_b'], bw['fc1_w'], bw['fc1_b'],
                                                        bw['fc2_wt'], bw['fc1_wt'], dn2w, dn2b, g.eps)
            else:
                def fc2_wgrad_fn(dx=dx, s=s, bp=bp):
                    ops.linear_wgrad(dx, s['hg'], G(bp + 'S2'), gelu_in=s['hpre'], colsum_out=G(bp + 'cs2'))
                side.run(fc2_wgrad_fn, dx, s['hg'])
                fused = use_fused_mlp(dt, C, 'bwd')
                if fused:       # fc2 dgrad * gp, fc1 dgrad and LayerNorm-2 backward (+ residual) in one kernel
                    dhd, dxmid = ops.mlp_bwd_dgrad(dx, s['hgp'], s['xmid'], bw['n2_w'], bw['fc2_wt'], bw['fc1_wt'], dn2w,
                                                   dn2b, g.eps)
                elif s['hpre']:
                    dhd = ops.linear_dgrad(dx, bw['fc2_wt'], gelu_pre=s['hg'])
                else:
                    dhd = ops.linear_dgrad(dx, bw['fc2_wt'], mul=s['hgp'])
                def fc1_wgrad_fn(dhd=dhd, s=s, bw=bw, bp=bp):
                    v2 = s['v2'] if s['v2'] is not None else ops.layernorm_fwd(s['xmid'], bw['n2_w'], bw['n2_b'], g.eps)
                    ops.linear_wgrad(dhd, v2, G(bp + 'mlp.net.0.0.weight'), colsum_out=G(bp + 'mlp.net.0.0.bias'))
                side.run(fc1_wgrad_fn, dhd, s['xmid'])
                if not fused:
                    if ops.linear_dgrad_ln_supported(dt, C, 4 * C):     # fc1 input gradient + norm2 backward + residual: one launch
                        dxmid = ops.linear_dgrad_ln(dhd, bw['fc1_w'], s['xmid'], dx, bw['n2_w'], dn2w, dn2b, g.eps)
                    else:
                        dv2 = ops.linear_dgrad(dhd, bw['fc1_wt'])
                        dxmid = ops.layernorm_bwd(s['xmid'], bw['n2_w'], dv2, dx, dn2w, dn2b, g.eps)
                        del dv2
                del dhd
            # attention branch: xmid = xin + g1 * (a Wp^T + bp)
            def proj_wgrad_fn(dxmid=dxmid, s=s, bp=bp):
                ops.linear_wgrad(dxmid, s['a'], G(bp + 'S1'), colsum_out=G(bp + 'cs1'))
            side.run(proj_wgrad_fn, dxmid, s['a'])
            if s['qkv'] is None:
                # fused: proj / attention / qkv input gradients and the norm1 backward in one launch (from the block input)
                has_n1 = bw['n1_w'] is not None
                if bi_flat == 0 and not has_n1 and sv.mask is None and tuning.get('route_attn_preln') != 0:
                    # the stage's first block: its input is the down-sampling norm's output, and nothing sits between them -
                    # the same launch carries the gradient through that norm (dy0 instead of dx)
                    dy0_fused, dqkv = ops.attn_block_bwd_preln(s['xin'], sv.y0, dxmid, sw.ln_w, bw['qkv_w'], bw['qkv_b'], bw['proj_wt'],
                                                               G(pre + 'downsample_cf2cl.norm.weight'),
                                                               G(pre + 'downsample_cf2cl.norm.bias'), F_, H, W, C, g.dim_head,
                                                               g.ph, g.pw, window, g.eps)
                    dx, u = None, None
                else:
                    dy0_fused = None
                    dx, dqkv, u = ops.attn_block_bwd(s['xin'], dxmid, bw['n1_w'], bw['n1_b'], bw['qkv_w'], bw['qkv_b'], bw['proj_wt'],
                                                     G(bp + 'norm1.weight') if has_n1 else None,
                                                     G(bp + 'norm1.bias') if has_n1 else None, F_, H, W, C, g.dim_head, g.ph, g.pw,
                                                     window, g.eps)
                u = s['xin'] if u is None else u
                def qkv_wgrad_fused_fn(dqkv=dqkv, u=u, bp=bp):
                    ops.linear_wgrad(dqkv, u, G(bp + 'self_attn.qkv.weight'), colsum_out=G(bp + 'self_attn.qkv.bias'))
                side.run(qkv_wgrad_fused_fn, dqkv, u)
                del dqkv, dxmid, u
                continue
            da = ops.linear_dgrad(dxmid, bw['proj_wt'])
            dqkv = ops.attn_bwd(s['qkv'], da, F_, H, W, C, g.dim_head, g.ph, g.pw, window)
            del da
            def qkv_wgrad_fn(dqkv=dqkv, s=s, bp=bp):
                ops.linear_wgrad(dqkv, s['u'], G(bp + 'self_attn.qkv.weight'), colsum_out=G(bp + 'self_attn.qkv.bias'))
            side.run(qkv_wgrad_fn, dqkv, s['xin'])
            if bw['n1_w'] is None and bi_flat == 0 and sv.mask is None and tuning.get('route_attn_preln') != 0 and \
                    ops.linear_dgrad_ln_supported(dt, C, 3 * C):
                # the stage's first block: qkv input gradient + residual cotangent carried through the down-sampling norm (one launch)
                dy0_fused = ops.linear_dgrad_preln(dqkv, bw['qkv_w'], sv.y0, dxmid, sw.ln_w, G(pre + 'downsample_cf2cl.norm.weight'),
                                                   G(pre + 'downsample_cf2cl.norm.bias'), g.eps)
                dx = None
            elif bw['n1_w'] is None:
                dx = ops.linear_dgrad(dqkv, bw['qkv_wt'], add=dxmid)
            elif ops.linear_dgrad_ln_supported(dt, C, 3 * C):           # qkv input gradient + norm1 backward + residual: one launch
                dx = ops.linear_dgrad_ln(dqkv, bw['qkv_w'], s['xin'], dxmid, bw['n1_w'], G(bp + 'norm1.weight'),
                                         G(bp + 'norm1.bias'), g.eps)
            else:
                du = ops.linear_dgrad(dqkv, bw['qkv_wt'])
                dx = ops.layernorm_bwd(s['xin'], bw['n1_w'], du, dxmid, G(bp + 'norm1.weight'), G(bp + 'norm1.bias'), g.eps)
                del du
            del dqkv, dxmid

    # ---- token mask, down-sampling LayerNorm + conv ---------------------------------------------------------
    if sv.mask is not None:
        ops.token_mask_bwd(dx, sv.mask, G(pre + 'mask_token').view(C))            # also zeroes dx on masked tokens
    if dy0_fused is not None:
        dy0 = dy0_fused                                                           # (rvt_attn_block_bwd_preln of the first block)
    else:
        dy0 = ops.layernorm_bwd(sv.y0, sw.ln_w, dx, None, G(pre + 'downsample_cf2cl.norm.weight'),
                                G(pre + 'downsample_cf2cl.norm.bias'), g.eps)
    def conv_wgrad_fn(dy0=dy0):
        if sv.inp.dtype == torch.uint8:
            ops.stem_wgrad(sv.inp, dy0, G('raw/conv'), g.H_in, g.W_in)
        else:
            ops.conv_wgrad(sv.inp, dy0, G('raw/conv'), g.k, g.stride, g.pad)
        if finalize is not None:
            finalize()           # in stream order behind every weight-gradient GEMM of this stage
    side.run(conv_wgrad_fn, sv.inp, dy0)
    d_in = None
    if need_input_grad:
        if sw.conv_wd4 is not None and tuning.get('route_conv_dgrad4') != 0 and \
                ops.conv_dgrad4_supported(dt, g.H_in, g.W_in, g.Cin, C, g.k, g.stride, g.pad, F_):
            d_in = ops.conv_dgrad4(dy0, sw.conv_wd4, prev_cot, g.H_in, g.W_in, g.Cin)      # one launch (2 x 2 pixel blocks)
        else:
            d_in = ops.conv_dgrad(dy0, sw.conv_wd, prev_cot, g.H_in, g.W_in, g.Cin, g.k, g.stride, g.pad)
    side.join()             # every parameter gradient of this stage is final from here on (DDP hook, optimizer)
    return d_in, dh0, dc0
