"""Kernel-side views of the module parameters (tiny host-side tensor shuffles, done once per step).

The nn.Module keeps the reference's parameter names and shapes; the HIP kernels want
  * conv weights tap-major / channels-last, input channels zero-padded to a multiple of 8,
  * transposed copies for the input-gradient GEMMs (LayerScale gamma folded in where it applies),
  * the ConvLSTM 1x1-conv rows interleaved so one 32-column epilogue unit holds all four gates,
  * everything cast to the activation dtype (fp32 masters stay in the module).
"""
from __future__ import annotations

from typing import List, Tuple

import torch

Tensor = torch.Tensor


def round8(c: int) -> int:
    return (c + 7) // 8 * 8


def pack_conv_fwd(w: Tensor, cin_pad: int, dtype: torch.dtype) -> Tensor:
    """[Cout,Cin,k,k] -> [Cout, k*k*cin_pad] (tap-major, cin fastest, zero padded channels)."""
    Cout, Cin, k, _ = w.shape
    t = w.permute(0, 2, 3, 1)
    if cin_pad != Cin:
        t = torch.nn.functional.pad(t, [0, cin_pad - Cin])
    return t.reshape(Cout, k * k * cin_pad).to(dtype).contiguous()


def unpack_conv_wgrad(dw: Tensor, Cin: int, k: int) -> Tensor:
    """[Cout, k*k*cin_pad] fp32 -> [Cout,Cin,k,k]."""
    Cout = dw.shape[0]
    cp = dw.shape[1] // (k * k)
    return dw.reshape(Cout, k, k, cp)[..., :Cin].permute(0, 3, 1, 2).contiguous()


def conv_dgrad_taps(k: int, stride: int, pad: int, parity: int) -> List[int]:
    return [t for t in range(k) if t % stride == (parity + pad) % stride]


def pack_conv_dgrad(w: Tensor, stride: int, pad: int, dtype: torch.dtype) -> Tensor:
    """Concatenation over parity classes (py,px) of [Cin][(a,b,cout)] = w[cout,cin,Ky[a],Kx[b]]
    — the layout rvt_conv_dgrad walks (rvt_amd/csrc/capi.hip)."""
    Cout, Cin, k, _ = w.shape
    parts = []
    for py in range(stride):
        for px in range(stride):
            ky = conv_dgrad_taps(k, stride, pad, py)
            kx = conv_dgrad_taps(k, stride, pad, px)
            sub = w[:, :, ky][:, :, :, kx]                       # (Cout,Cin,nky,nkx)
            parts.append(sub.permute(1, 2, 3, 0).reshape(-1))
    return torch.cat(parts).to(dtype).contiguous()


def lstm_gate_perm(C: int, device) -> Tensor:
    """index[n'] = original row (gate*C + c) placed at interleaved row n' = (c/8)*32 + gate*8 + c%8."""
    n = torch.arange(4 * C, device=device)
    c = (n // 32) * 8 + n % 8
    gate = (n % 32) // 8
    return gate * C + c


class StageWeights:
    """Everything one stage's kernels read, derived from the fp32 parameters."""

    def __init__(self, p: dict, pre: str, C: int, Cin: int, k: int, stride: int, pad: int, num_blocks: int,
                 dtype: torch.dtype, need_grad: bool):
        f32 = torch.float32
        g = lambda n: p[pre + n].detach()
        self.C, self.Cin, self.cin_pad = C, Cin, round8(Cin)
        wc = g('downsample_cf2cl.conv.weight').to(f32)
        self.conv_w = pack_conv_fwd(wc, self.cin_pad, dtype)
        self.conv_wd = pack_conv_dgrad(wc, stride, pad, dtype) if (need_grad and Cin % 8 == 0) else None
        self.ln_w = g('downsample_cf2cl.norm.weight').to(f32).contiguous()
        self.ln_b = g('downsample_cf2cl.norm.bias').to(f32).contiguous()
        self.blocks: List[Tuple[dict, dict]] = []
        for bi in range(num_blocks):
            pair = []
            for blk in ('att_window', 'att_grid'):
                bp = f'att_blocks.{bi}.{blk}.'
                d = {}
                has_n1 = (pre + bp + 'norm1.weight') in p
                d['n1_w'] = g(bp + 'norm1.weight').to(f32).contiguous() if has_n1 else None
                d['n1_b'] = g(bp + 'norm1.bias').to(f32).contiguous() if has_n1 else None
                wq, wp = g(bp + 'self_attn.qkv.weight').to(f32), g(bp + 'self_attn.proj.weight').to(f32)
                w1, w2 = g(bp + 'mlp.net.0.0.weight').to(f32), g(bp + 'mlp.net.2.weight').to(f32)
                g1, g2 = g(bp + 'ls1.gamma').to(f32).contiguous(), g(bp + 'ls2.gamma').to(f32).contiguous()
                d.update(qkv_w=wq.to(dtype).contiguous(), qkv_b=g(bp + 'self_attn.qkv.bias').to(f32).contiguous(),
                         proj_w=wp.to(dtype).contiguous(), proj_b=g(bp + 'self_attn.proj.bias').to(f32).contiguous(),
                         g1=g1, g2=g2,
                         n2_w=g(bp + 'norm2.weight').to(f32).contiguous(), n2_b=g(bp + 'norm2.bias').to(f32).contiguous(),
                         fc1_w=w1.to(dtype).contiguous(), fc1_b=g(bp + 'mlp.net.0.0.bias').to(f32).contiguous(),
                         fc2_w=w2.to(dtype).contiguous(), fc2_b=g(bp + 'mlp.net.2.bias').to(f32).contiguous())
                if need_grad:
                    # dgrad operands: W^T, with the LayerScale of the branch folded into proj / fc2
                    d.update(qkv_wt=wq.t().to(dtype).contiguous(),
                             proj_wt=(wp * g1[:, None]).t().to(dtype).contiguous(),
                             fc1_wt=w1.t().to(dtype).contiguous(),
                             fc2_wt=(w2 * g2[:, None]).t().to(dtype).contiguous(),
                             proj_w32=wp, fc2_w32=w2)
                pair.append(d)
            self.blocks.append(tuple(pair))
        # optional depth-wise 3x3 of the DWS-ConvLSTM (rnn.py:25-29): fp32 [Cg][k*k] + bias, Cg = C (hidden only) or 2C
        self.dws = None
        if (pre + 'lstm.conv3x3_dws.weight') in p:
            wd = g('lstm.conv3x3_dws.weight').to(f32)
            cg, kk = wd.shape[0], wd.shape[-1]
            self.dws = dict(only_hidden=(cg == C), k=kk, w=wd.reshape(cg, kk * kk).contiguous(),
                            b=g('lstm.conv3x3_dws.bias').to(f32).contiguous())
        wl = g('lstm.conv1x1.weight').to(f32).reshape(4 * C, 2 * C)
        perm = lstm_gate_perm(C, wl.device)
        self.lstm_perm = perm
        self.lstm_w = wl[perm].to(dtype).contiguous()
        self.lstm_b = g('lstm.conv1x1.bias').to(f32)[perm].contiguous()
        self.lstm_wt = wl.t().to(dtype).contiguous() if need_grad else None
