"""CPU oracle for the RVT recurrent MaxViT backbone hot path.

*** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  ``rvt_amd`` never imports it; the product
path runs on hand-written HIP kernels and raises if they are missing.

This is a from-scratch restatement, in plain channels-last tensor algebra on the
CPU (torch used as an ndarray + autograd library; fp32 or fp64), of the algorithm
the reference implements in

  /root/reference/models/detection/recurrent_backbone/maxvit_rnn.py   (RNNDetector, stages)
  /root/reference/models/layers/maxvit/maxvit.py                      (downsample, attention blocks)
  /root/reference/models/layers/rnn.py                                (DWSConvLSTM2d)
  /root/reference/modules/detection.py:131-148                        (time loop)

Each function cites the reference lines it follows.  Parameters are addressed by
the reference's own ``state_dict`` names (``stages.{i}.…``) so that a state dict
taken from the reference module can be fed in unchanged.

Parity pinning: the reference ships NO tests / golden vectors for this path
(SURVEY.md §4, §8c).  The oracle is pinned instead against outputs of the
unmodified reference imported in the authoring container
(``oracle/make_golden.py`` → ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
re-checks the oracle against those fixtures on every run).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Optional[Tuple[Tensor, Tensor]]


@dataclass
class OracleCfg:
    """The subset of ``model.backbone`` the hot path reads
    (reference: config/model/maxvit_yolox/default.yaml:6-42)."""
    input_channels: int = 20
    embed_dim: int = 64
    dim_multiplier: Tuple[int, ...] = (1, 2, 4, 8)
    num_blocks: Tuple[int, ...] = (1, 1, 1, 1)
    patch_size: int = 4                      # stem.patch_size
    partition_size: Tuple[int, int] = (6, 10)
    dim_head: int = 32
    overlap: bool = True                     # stage.downsample.overlap
    norm_eps: float = 1e-5
    dws_conv: bool = False
    dws_conv_only_hidden: bool = True
    dws_conv_kernel_size: int = 3
    enable_masking: bool = False
    # 'im2col': conv, LayerNorm and GELU stated as explicit arithmetic (unfold + contraction, mean / variance, erf) - what the parity
    # tests use; 'aten': the fused ATen ops the reference itself calls (F.conv2d maxvit.py:160-168, F.layer_norm layers/norm.py:44-56,
    # F.gelu layers/activations.py:138-145) - used by bench.py's cpu_baseline leg so that the timed port does not understate the
    # reference; tests/test_oracle_golden.py checks that both give the same numbers
    conv_impl: str = 'im2col'

    @property
    def stage_dims(self) -> List[int]:
        return [self.embed_dim * m for m in self.dim_multiplier]


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
_ATEN = [False]          # set by sequence_forward from OracleCfg.conv_impl == 'aten' (timing mode, see OracleCfg)


def layer_norm_last(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """LayerNorm over the last (channel) axis, biased variance.
    reference: maxvit.py:172,177 (`LayerNorm(num_channels, eps=1e-5)` → F.layer_norm,
    layers/norm.py:44-56)."""
    if _ATEN[0]:
        return F.layer_norm(x, (x.shape[-1],), w, b, eps)
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """Exact (erf) GELU.  reference: layers/activations.py:138-145 (F.gelu)."""
    if _ATEN[0]:
        return F.gelu(x)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def conv_downsample_ln(x_nchw: Tensor, p: Dict[str, Tensor], pre: str, factor: int,
                       overlap: bool, eps: float, conv_impl: str = 'im2col') -> Tensor:
    """Strided conv (bias-free) → channels-last → LayerNorm.
    reference: maxvit.py:143-178 — kernel (f-1)*2+1, padding k//2, stride f when
    ``overlap`` else kernel f, padding 0."""
    w = p[pre + 'conv.weight']
    k = (factor - 1) * 2 + 1 if overlap else factor
    pad = k // 2 if overlap else 0
    assert w.shape[-1] == k and w.shape[-2] == k
    # explicit im2col contraction (not F.conv2d) so the oracle states the arithmetic itself
    B, Cin, H, W = x_nchw.shape
    Ho = (H + 2 * pad - k) // factor + 1
    Wo = (W + 2 * pad - k) // factor + 1
    if conv_impl == 'aten':
        y = F.conv2d(x_nchw, w, None, stride=factor, padding=pad).permute(0, 2, 3, 1)
    else:
        cols = F.unfold(x_nchw, kernel_size=k, padding=pad, stride=factor)     # (B, Cin*k*k, Ho*Wo)
        y = torch.einsum('bkn,ok->bno', cols, w.reshape(w.shape[0], -1))       # (B, N, Cout)
        y = y.reshape(B, Ho, Wo, -1)
    return layer_norm_last(y, p[pre + 'norm.weight'], p[pre + 'norm.bias'], eps)


def partition_indices(H: int, W: int, ph: int, pw: int, window: bool) -> Tensor:
    """Token index map ``idx[p, l]`` = flat (y*W+x) position of slot ``l`` of partition ``p``.

    window (maxvit.py:273-279): partition (y//ph, x//pw), slot (y%ph)*pw + (x%pw).
    grid   (maxvit.py:290-296): with G=(ph,pw) the grid size, partition (y % (H/ph), x % (W/pw)),
                                slot (y // (H/ph))*pw + (x // (W/pw))  — dilated sampling.
    """
    assert H % ph == 0 and W % pw == 0
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    if window:
        part = (ys // ph) * (W // pw) + (xs // pw)
        slot = (ys % ph) * pw + (xs % pw)
    else:
        sh, sw = H // ph, W // pw
        part = (ys % sh) * sw + (xs % sw)
        slot = (ys // sh) * pw + (xs // sw)
    idx = torch.empty((H // ph) * (W // pw), ph * pw, dtype=torch.long)
    idx[part.reshape(-1), slot.reshape(-1)] = torch.arange(H * W)
    return idx


def self_attention_partitioned(x: Tensor, p: Dict[str, Tensor], pre: str, ph: int, pw: int,
                               window: bool, dim_head: int) -> Tensor:
    """Multi-head self attention inside each partition; x is (B,H,W,C), returns same shape.
    reference: maxvit.py:252-265 (partition → attn → reverse) and 343-354:
    qkv linear, per-head channel layout [q(dh) k(dh) v(dh)], softmax(q·kᵀ·dh^-0.5)·v,
    heads concatenated (channel = head*dh + d), proj linear.  No bias/mask."""
    B, H, W, C = x.shape
    heads = C // dim_head
    idx = partition_indices(H, W, ph, pw, window)           # (P, L)
    P, L = idx.shape
    tok = x.reshape(B, H * W, C)[:, idx]                      # (B, P, L, C)
    qkv = tok @ p[pre + 'qkv.weight'].t() + p[pre + 'qkv.bias']
    qkv = qkv.reshape(B, P, L, heads, 3, dim_head)
    q, k, v = qkv[..., 0, :], qkv[..., 1, :], qkv[..., 2, :]  # (B,P,L,heads,dh)
    s = torch.einsum('bplhd,bpmhd->bphlm', q, k) * (dim_head ** -0.5)
    a = torch.softmax(s, dim=-1)
    o = torch.einsum('bphlm,bpmhd->bplhd', a, v).reshape(B, P, L, C)
    o = o @ p[pre + 'proj.weight'].t() + p[pre + 'proj.bias']
    out = torch.empty_like(o).reshape(B, H * W, C)
    out = out.index_copy(1, idx.reshape(-1), o.reshape(B, P * L, C))
    return out.reshape(B, H, W, C)


def attention_block(x: Tensor, p: Dict[str, Tensor], pre: str, ph: int, pw: int, window: bool,
                    dim_head: int, eps: float) -> Tensor:
    """x += γ1·Attn(LN1(x));  x += γ2·MLP(LN2(x)).
    reference: maxvit.py:267-270; norm1 is Identity for the first window block of a stage
    (maxvit_rnn.py:153, maxvit.py:234) — detected here by the absence of its parameters;
    MLP = Linear(C,4C) → GELU → Linear(4C,C) (maxvit.py:100-118, non-gated)."""
    u = x
    if pre + 'norm1.weight' in p:
        u = layer_norm_last(x, p[pre + 'norm1.weight'], p[pre + 'norm1.bias'], eps)
    x = x + p[pre + 'ls1.gamma'] * self_attention_partitioned(u, p, pre + 'self_attn.', ph, pw, window, dim_head)
    v = layer_norm_last(x, p[pre + 'norm2.weight'], p[pre + 'norm2.bias'], eps)
    hid = gelu_erf(v @ p[pre + 'mlp.net.0.0.weight'].t() + p[pre + 'mlp.net.0.0.bias'])
    x = x + p[pre + 'ls2.gamma'] * (hid @ p[pre + 'mlp.net.2.weight'].t() + p[pre + 'mlp.net.2.bias'])
    return x


def conv_lstm(x: Tensor, state: State, p: Dict[str, Tensor], pre: str, cfg: OracleCfg) -> Tuple[Tensor, Tensor]:
    """(DWS-)ConvLSTM cell on channels-last maps; x, h, c are (B,H,W,C).
    reference: rnn.py:36-69 — zero state when None (:43-47); optional depth-wise 3x3 on h
    (:50-51) or on cat(x,h) (:53-54); 1x1 conv on cat(x,h) (:52,55); output channel blocks
    [f, i, o | g] (:57-64); c = f*c + i*g (:66); h = o*tanh(c) (:67)."""
    C = x.shape[-1]
    if state is None:
        h = torch.zeros_like(x)
        c = torch.zeros_like(x)
    else:
        h, c = state

    def dw(t: Tensor) -> Tensor:
        w = p[pre + 'conv3x3_dws.weight']                # (Cg,1,k,k)
        b = p[pre + 'conv3x3_dws.bias']
        k = w.shape[-1]
        y = F.conv2d(t.permute(0, 3, 1, 2), w, b, padding=k // 2, groups=w.shape[0])
        return y.permute(0, 2, 3, 1)

    if cfg.dws_conv and cfg.dws_conv_only_hidden:
        h = dw(h)
    xh = torch.cat((x, h), dim=-1)
    if cfg.dws_conv and not cfg.dws_conv_only_hidden:
        xh = dw(xh)
    w = p[pre + 'conv1x1.weight'].reshape(4 * C, 2 * C)
    mix = xh @ w.t() + p[pre + 'conv1x1.bias']
    f = torch.sigmoid(mix[..., 0 * C:1 * C])
    i = torch.sigmoid(mix[..., 1 * C:2 * C])
    o = torch.sigmoid(mix[..., 2 * C:3 * C])
    g = torch.tanh(mix[..., 3 * C:4 * C])
    c_new = f * c + i * g
    h_new = o * torch.tanh(c_new)
    return h_new, c_new


# ----------------------------------------------------------------------------
# stage / backbone / sequence
# ----------------------------------------------------------------------------

def stage_forward(x_nchw: Tensor, state: State, token_mask: Optional[Tensor], p: Dict[str, Tensor],
                  si: int, cfg: OracleCfg) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    """One stage.  States and the returned feature are channels-last (B,H,W,C) here.
    reference: maxvit_rnn.py:169-182."""
    pre = f'stages.{si}.'
    factor = cfg.patch_size if si == 0 else 2
    x = conv_downsample_ln(x_nchw, p, pre + 'downsample_cf2cl.', factor, cfg.overlap, cfg.norm_eps, cfg.conv_impl)
    if token_mask is not None:
        # maxvit_rnn.py:174-176: x[token_mask] = mask_token
        x = torch.where(token_mask[..., None], p[pre + 'mask_token'].reshape(1, 1, 1, -1).to(x.dtype), x)
    ph, pw = cfg.partition_size
    for bi in range(cfg.num_blocks[si]):
        bpre = f'{pre}att_blocks.{bi}.'
        x = attention_block(x, p, bpre + 'att_window.', ph, pw, True, cfg.dim_head, cfg.norm_eps)
        x = attention_block(x, p, bpre + 'att_grid.', ph, pw, False, cfg.dim_head, cfg.norm_eps)
    h, c = conv_lstm(x, state, p, pre + 'lstm.', cfg)
    return h, (h, c)


def backbone_forward(x_nchw: Tensor, prev_states: Optional[Sequence[State]], token_mask: Optional[Tensor],
                     p: Dict[str, Tensor], cfg: OracleCfg):
    """One time step through the four stages.  Returns ({1..4: (B,C,H,W)}, [(h,c)]*4) with
    NCHW-shaped features/states like the reference (maxvit_rnn.py:93-105)."""
    if prev_states is None:
        prev_states = [None] * 4
    feats, states = {}, []
    x = x_nchw
    for si in range(4):
        st = prev_states[si]
        st_cl = None if st is None else (st[0].permute(0, 2, 3, 1), st[1].permute(0, 2, 3, 1))
        h, (h_, c) = stage_forward(x, st_cl, token_mask if si == 0 else None, p, si, cfg)
        x = h.permute(0, 3, 1, 2)
        feats[si + 1] = x
        states.append((x, c.permute(0, 3, 1, 2)))
    return feats, states


def pad_to(x: Tensor, hw: Tuple[int, int]) -> Tensor:
    """Zero-pad bottom/right to the model resolution.
    reference: utils/padding.py:29-44 (`type='corner'`), modules/detection.py:134."""
    H, W = x.shape[-2:]
    return F.pad(x, [0, hw[1] - W, 0, hw[0] - H])


def sequence_forward(xs: Tensor, prev_states, p: Dict[str, Tensor], cfg: OracleCfg,
                     in_res_hw: Tuple[int, int], dtype=torch.float32,
                     token_masks: Optional[Tensor] = None):
    """The time loop: xs is (T,B,Cin,h,w) (uint8 or float); cast, pad, step.
    reference: modules/detection.py:131-148.
    Returns ([{stage: (B,C,H,W)} for t in 0..T-1], final_states)."""
    outs = []
    states = prev_states
    _ATEN[0] = cfg.conv_impl == 'aten'
    for t in range(xs.shape[0]):
        x = pad_to(xs[t].to(dtype), in_res_hw)
        tm = None if token_masks is None else token_masks[t]
        feats, states = backbone_forward(x, states, tm, p, cfg)
        outs.append(feats)
    return outs, states


def reset_states(states, is_first_sample: Tensor):
    """Zero the state rows of samples that start a new sequence.
    reference: modules/utils/detection.py:96-113 (`inp[indices_or_bool_tensor] = 0`)."""
    if states is None:
        return None
    keep = (~is_first_sample).to(states[0][0].dtype).reshape(-1, 1, 1, 1)
    return [(h * keep, c * keep) for h, c in states]
