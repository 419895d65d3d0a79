"""Tensor-level wrappers over the C ABI (include/rvt_hip.h).  torch is used for device memory and
streams only; every op below is one HIP kernel launch (conv_dgrad: one per stride-parity class)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor


def _out(like: Tensor, shape, dtype=None, out: Optional[Tensor] = None) -> Tensor:
    if out is not None:
        assert tuple(out.shape) == tuple(shape) and out.is_contiguous()
        return out
    return torch.empty(shape, dtype=dtype or like.dtype, device=like.device)


_WS = {}


def _wgrad_ws(like: Tensor, out_rows: int, out_cols: int, tokens: int, want_colsum: bool) -> Tensor:
    """Grow-only scratch for the two-stage split-K reduction, one per (device, stream): kernels on one stream serialise,
    so consecutive weight-gradient launches of that stream can share it; launches on different streams never do."""
    n = L.get_lib().rvt_wgrad_workspace_floats(L.dtype_code(like.dtype), out_rows, out_cols, tokens, int(want_colsum))
    st = L.stream_of(like)
    key = (like.device.type, like.device.index, 0 if st is None else int(st))
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=like.device)
        _WS[key] = ws
    return ws


def prepack_input(src: Tensor, H: int, W: int, Cp: int, dtype: torch.dtype, out: Optional[Tensor] = None) -> Tensor:
    """(F,Cin,h,w) uint8/float32 -> (F,H,W,Cp) `dtype`, zero padded (cast+pad of modules/detection.py:133-134)."""
    assert src.dim() == 4 and src.dtype in (torch.uint8, torch.float32)
    src = src.contiguous()
    F_, Cin, h, w = src.shape
    dst = _out(src, (F_, H, W, Cp), dtype, out)
    L.call('rvt_prepack_input', L.ptr(src), int(src.dtype == torch.uint8), L.ptr(dst), L.dtype_code(dtype),
           F_, Cin, h, w, H, W, Cp, L.stream_of(src))
    return dst


def conv_out_hw(H: int, W: int, k: int, stride: int, pad: int):
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def conv_fwd(x: Tensor, w: Tensor, k: int, stride: int, pad: int, out: Optional[Tensor] = None) -> Tensor:
    F_, H, W, Cin = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == k * k * Cin and w.dtype == x.dtype
    Ho, Wo = conv_out_hw(H, W, k, stride, pad)
    y = _out(x, (F_, Ho, Wo, Cout), out=out)
    L.call('rvt_conv_fwd', L.ptr(x), L.ptr(w), L.ptr(y), L.dtype_code(x.dtype), F_, H, W, Cin, Cout, k, stride, pad,
           L.stream_of(x))
    return y


def conv_bn_act_fwd(x: Tensor, w: Tensor, scale: Tensor, shift: Tensor, k: int, stride: int, pad: int, act: int = 1) -> Tensor:
    """Inference-mode conv + BatchNorm affine + activation in one launch (scale / shift fp32 [Cout])."""
    F_, H, W, Cin = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == k * k * Cin and w.dtype == x.dtype and scale.dtype == torch.float32 and shift.dtype == torch.float32
    Ho, Wo = conv_out_hw(H, W, k, stride, pad)
    y = _out(x, (F_, Ho, Wo, Cout))
    L.call('rvt_conv_bn_act_fwd', L.ptr(x), L.ptr(w), L.ptr(scale), L.ptr(shift), L.ptr(y), L.dtype_code(x.dtype), F_, H, W, Cin, Cout,
           k, stride, pad, act, L.stream_of(x))
    return y


def stem_supported(src: Tensor, dtype: torch.dtype, Cout: int, k: int, stride: int, pad: int) -> bool:
    """The stem kernels (csrc/stem.hpp) take the loader's uint8 planes; everything else goes prepack + conv + LayerNorm."""
    return bool(L.get_lib().rvt_stem_supported(L.dtype_code(dtype), int(src.dtype == torch.uint8), src.shape[1], Cout, k, stride,
                                               pad, src.shape[3]))


def stem_fwd(src: Tensor, w: Tensor, ln_w: Tensor, ln_b: Tensor, H: int, W: int, eps: float):
    """src (F,Cin,h,w) uint8, w = packed conv weight (64, 49*cp) -> y0 = conv(pad(src)), x = LayerNorm(y0), (F,Ho,Wo,64)."""
    assert src.dim() == 4 and src.dtype == torch.uint8 and src.is_contiguous()
    F_, Cin, h, wd = src.shape
    cp = w.shape[1] // 49
    assert w.shape[0] == 64 and w.shape[1] == 49 * cp and cp >= Cin
    Ho, Wo = conv_out_hw(H, W, 7, 4, 3)
    y0 = torch.empty((F_, Ho, Wo, 64), dtype=w.dtype, device=src.device)
    x = torch.empty_like(y0)
    L.call('rvt_stem_fwd', L.ptr(src), L.ptr(w), L.ptr(ln_w), L.ptr(ln_b), L.ptr(y0), L.ptr(x), L.dtype_code(w.dtype), F_, Cin, cp,
           h, wd, H, W, float(eps), L.stream_of(y0))
    return y0, x


def stem_wgrad(src: Tensor, dy: Tensor, dw: Tensor, H: int, W: int) -> None:
    """dw (64, 49*cp) fp32 += dy^T im2col(pad(src)) — the layout conv_wgrad writes."""
    assert src.dtype == torch.uint8 and src.is_contiguous() and dy.is_contiguous() and dw.dtype == torch.float32
    F_, Cin, h, wd = src.shape
    cp = dw.shape[1] // 49
    assert tuple(dw.shape) == (64, 49 * cp) and tuple(dy.shape[:1]) == (F_,) and dy.shape[-1] == 64
    n = L.get_lib().rvt_stem_wgrad_ws_floats(Cin, F_, H, W)
    st = L.stream_of(dy)
    key = ('stem', dy.device.type, dy.device.index, 0 if st is None else int(st))
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.float32, device=dy.device)
        _WS[key] = ws
    L.call('rvt_stem_wgrad', L.ptr(src), L.ptr(dy), L.ptr(dw), L.ptr(ws), L.dtype_code(dy.dtype), F_, Cin, cp, h, wd, H, W, st)


def conv_dgrad4_supported(dtype: torch.dtype, H: int, W: int, Cin: int, Cout: int, k: int, stride: int, pad: int, F_: int) -> bool:
    return dtype in L._DT and bool(L.get_lib().rvt_conv_dgrad4_supported(L.dtype_code(dtype), H, W, Cin, Cout, k, stride, pad, F_))


def conv_dgrad4(dy: Tensor, wd4: Tensor, add: Optional[Tensor], H: int, W: int, Cin: int) -> Tensor:
    """Input gradient of the 3x3 / stride-2 / pad-1 conv in one launch (2x2 input-pixel blocks; wd4 = weights.pack PACK_CONV_DGRAD4)."""
    F_, Ho, Wo, Cout = dy.shape
    assert tuple(wd4.shape) == (4 * Cin, 4 * Cout) and H == 2 * Ho and W == 2 * Wo
    din = torch.empty((F_, H, W, Cin), dtype=dy.dtype, device=dy.device)
    L.call('rvt_conv_dgrad4', L.ptr(dy), L.ptr(wd4), L.ptr(add), L.ptr(din), L.dtype_code(dy.dtype), F_, H, W, Cin, Cout,
           L.stream_of(dy))
    return din


def conv_dgrad(dy: Tensor, wd: Tensor, add: Optional[Tensor], H: int, W: int, Cin: int, k: int, stride: int, pad: int,
               out: Optional[Tensor] = None) -> Tensor:
    F_, Ho, Wo, Cout = dy.shape
    din = _out(dy, (F_, H, W, Cin), out=out)
    L.call('rvt_conv_dgrad', L.ptr(dy), L.ptr(wd), L.ptr(add), L.ptr(din), L.dtype_code(dy.dtype), F_, H, W, Cin, Cout,
           k, stride, pad, L.stream_of(dy))
    return din


def conv_wgrad(x: Tensor, dy: Tensor, dw: Tensor, k: int, stride: int, pad: int) -> None:
    F_, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, k * k * Cin)
    ws = _wgrad_ws(x, Cout, k * k * Cin, dy.numel() // Cout, False)
    L.call('rvt_conv_wgrad', L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), L.dtype_code(x.dtype), F_, H, W, Cin, Cout, k,
           stride, pad, L.stream_of(x))


def layernorm_fwd(x: Tensor, w: Tensor, b: Tensor, eps: float, out: Optional[Tensor] = None) -> Tensor:
    C = x.shape[-1]
    rows = x.numel() // C
    y = _out(x, x.shape, out=out)
    L.call('rvt_layernorm_fwd', L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.dtype_code(x.dtype), rows, C, float(eps),
           L.stream_of(x))
    return y


def layernorm_bwd(x: Tensor, w: Tensor, dy: Tensor, dres: Optional[Tensor], dw: Tensor, db: Tensor, eps: float,
                  out: Optional[Tensor] = None) -> Tensor:
    C = x.shape[-1]
    rows = x.numel() // C
    dx = _out(x, x.shape, out=out)
    L.call('rvt_layernorm_bwd', L.ptr(x), L.ptr(w), L.ptr(dy), L.ptr(dres), L.ptr(dx), L.ptr(dw), L.ptr(db),
           L.dtype_code(x.dtype), rows, C, float(eps), L.stream_of(x))
    return dx


def ln_linear_supported(dtype: torch.dtype, C: int, N: int) -> bool:
    return dtype in L._DT and bool(L.get_lib().rvt_ln_linear_supported(L.dtype_code(dtype), C, N))


def ln_linear_fwd(x: Tensor, ln_w: Tensor, ln_b: Tensor, w: Tensor, bias: Optional[Tensor], eps: float, want_u: bool = True):
    """(u, y) = (LN(x), LN(x) @ w.T + bias) in one launch (csrc/ln_linear.hpp; reference maxvit.py:268 -> :347, `qkv(norm1(x))`);
    u is None when want_u is False (no-grad forward) or when there is no LayerNorm (ln_w = ln_b = None: y = x @ w.T + bias)."""
    C = x.shape[-1]
    N = w.shape[0]
    rows = x.numel() // C
    u = _out(x, x.shape) if want_u and ln_w is not None else None
    y = _out(x, x.shape[:-1] + (N,))
    L.call('rvt_ln_linear_fwd', L.ptr(x), L.ptr(ln_w), L.ptr(ln_b), L.ptr(w), L.ptr(bias), L.ptr(u), L.ptr(y),
           L.dtype_code(x.dtype), rows, C, N, float(eps), L.stream_of(x))
    return u, y


def linear_dgrad_ln_supported(dtype: torch.dtype, C: int, K: int) -> bool:
    return dtype in L._DT and bool(L.get_lib().rvt_linear_dgrad_ln_supported(L.dtype_code(dtype), C, K))


def linear_dgrad_ln(dy: Tensor, w: Tensor, x: Tensor, dres: Optional[Tensor], ln_w: Tensor, dw: Tensor, db: Tensor,
                    eps: float, out: Optional[Tensor] = None) -> Tensor:
    """dx = dres + LN'(dy @ w; x) with dw += sum (dy @ w) * xhat, db += sum (dy @ w): the input gradient of a linear layer fed by
    a LayerNorm and that LayerNorm's backward in one launch (csrc/dgrad_ln.hpp).  w: the forward weight [K][C]."""
    C = x.shape[-1]
    K = dy.shape[-1]
    rows = x.numel() // C
    assert tuple(w.shape) == (K, C) and w.dtype == x.dtype == dy.dtype and dy.numel() // K == rows
    dx = _out(x, x.shape, out=out)
    L.call('rvt_linear_dgrad_ln', L.ptr(dy), L.ptr(w), L.ptr(x), L.ptr(dres), L.ptr(dx), L.ptr(ln_w), L.ptr(dw), L.ptr(db),
           L.dtype_code(x.dtype), rows, C, K, float(eps), L.stream_of(x))
    return dx


def linear_dgrad_preln(dy: Tensor, w: Tensor, y0: Tensor, add: Tensor, ln_w: Tensor, dw: Tensor, db: Tensor, eps: float) -> Tensor:
    """dy0 = LN'(dy @ w + add; y0): the qkv input gradient of a stage's first block (no norm1) plus its residual cotangent, carried
    through the down-sampling norm in front of the block (maxvit.py:177) in one launch; dw / db += that norm's parameter gradients."""
    C = y0.shape[-1]
    K = dy.shape[-1]
    rows = y0.numel() // C
    assert tuple(w.shape) == (K, C) and w.dtype == y0.dtype == dy.dtype == add.dtype and dy.numel() // K == rows
    out = torch.empty_like(y0)
    L.call('rvt_linear_dgrad_preln', L.ptr(dy), L.ptr(w), L.ptr(y0), L.ptr(add), L.ptr(out), L.ptr(ln_w), L.ptr(dw), L.ptr(db),
           L.dtype_code(y0.dtype), rows, C, K, float(eps), L.stream_of(y0))
    return out


def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], gelu_in: bool = False, out: Optional[Tensor] = None) -> Tensor:
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype
    y = _out(x, (*x.shape[:-1], N), out=out)
    L.call('rvt_linear_fwd', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), L.dtype_code(x.dtype), M, N, K, int(gelu_in),
           L.stream_of(x))
    return y


def linear_scale_res_fwd(x: Tensor, w: Tensor, bias: Tensor, gamma: Tensor, res: Tensor, gelu_in: bool = False,
                         out: Optional[Tensor] = None) -> Tensor:
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K and res.shape[-1] == N
    y = _out(res, res.shape, out=out)
    L.call('rvt_linear_scale_res_fwd', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(gamma), L.ptr(res), L.ptr(y),
           L.dtype_code(x.dtype), M, N, K, int(gelu_in), L.stream_of(x))
    return y


def linear_gelu_fwd(x: Tensor, w: Tensor, bias: Tensor, want_grad: bool):
    """g = GELU(x W^T + b), gp = GELU'(x W^T + b) (None unless want_grad)."""
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    g = torch.empty((*x.shape[:-1], N), dtype=x.dtype, device=x.device)
    gp = torch.empty_like(g) if want_grad else None
    L.call('rvt_linear_gelu_fwd', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(g), L.ptr(gp), L.dtype_code(x.dtype), M, N, K,
           L.stream_of(x))
    return g, gp


def mlp_fused_supported(dtype: torch.dtype, C: int) -> bool:
    return bool(L.get_lib().rvt_mlp_fused_supported(L.dtype_code(dtype), C))


def mlp_fwd(xmid: Tensor, ln_w: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, gamma: Tensor,
            eps: float, want_grad: bool = False, out: Optional[Tensor] = None, want_v2: bool = False, want_pre: bool = False):
    """xout = xmid + gamma*(GELU(LN(xmid) W1^T + b1) W2^T + b2) in one fused kernel; with want_grad also returns
    g = GELU(h), gp = GELU'(h) (the only intermediates backward needs); with want_pre instead the pre-activation h alone
    (returned in g's place, gp None).  Returns (xout, g, gp), or with want_v2 (xout, g, gp, LN(xmid))."""
    C = xmid.shape[-1]
    M = xmid.numel() // C
    y = _out(xmid, xmid.shape, out=out)
    g = gp = None
    if want_grad or want_pre:
        g = torch.empty((*xmid.shape[:-1], 4 * C), dtype=xmid.dtype, device=xmid.device)
        gp = None if want_pre else torch.empty_like(g)
    v2 = torch.empty_like(xmid) if want_v2 else None
    L.call('rvt_mlp_fwd', L.ptr(xmid), L.ptr(y), L.ptr(g), L.ptr(gp), L.ptr(v2), L.ptr(ln_w), L.ptr(ln_b), L.ptr(w1),
           L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(gamma), L.dtype_code(xmid.dtype), M, C, float(eps), L.stream_of(xmid))
    return (y, g, gp, v2) if want_v2 else (y, g, gp)


def mlp_bwd_dgrad(dxout: Tensor, gp: Tensor, xmid: Tensor, ln_w: Tensor, w2g_t: Tensor, w1_t: Tensor, dln_w: Tensor,
                  dln_b: Tensor, eps: float):
    """dh = (dxout (W2*gamma)) * gp;  dxmid = dxout + LN'(dh W1; xmid);  dln_w/dln_b += …  Returns (dh, dxmid)."""
    C = xmid.shape[-1]
    M = xmid.numel() // C
    dh = torch.empty_like(gp)
    dxmid = torch.empty_like(xmid)
    L.call('rvt_mlp_bwd_dgrad', L.ptr(dxout), L.ptr(gp), L.ptr(xmid), L.ptr(dh), L.ptr(dxmid), L.ptr(ln_w), L.ptr(w2g_t),
           L.ptr(w1_t), L.ptr(dln_w), L.ptr(dln_b), L.dtype_code(xmid.dtype), M, C, float(eps), L.stream_of(xmid))
    return dh, dxmid


def mlp_bwd_fused_supported(dtype: torch.dtype, C: int) -> bool:
    return bool(L.get_lib().rvt_mlp_bwd_fused_supported(L.dtype_code(dtype), C))


def _mlp_ws(xmid: Tensor) -> Tensor:
    C = xmid.shape[-1]
    M = xmid.numel() // C
    n = L.get_lib().rvt_mlp_bwd_fused_ws_floats(L.dtype_code(xmid.dtype), C, M)
    st = L.stream_of(xmid)
    key = ('mlpbwd', xmid.device.type, xmid.device.index, 0 if st is None else int(st))
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.float32, device=xmid.device)
        _WS[key] = ws
    return ws


def mlp_bwd_recompute_dgrad(dxout: Tensor, xmid: Tensor, ln_w: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2g_t: Tensor,
                            w1_t: Tensor, dln_w: Tensor, dln_b: Tensor, eps: float) -> Tensor:
    """Input-gradient half of the recompute backward: dxmid = dxout + LN2'(...); dln_w / dln_b += ."""
    C = xmid.shape[-1]
    M = xmid.numel() // C
    dxmid = torch.empty_like(xmid)
    L.call('rvt_mlp_bwd_recompute_dgrad', L.ptr(dxout), L.ptr(xmid), L.ptr(dxmid), L.ptr(ln_w), L.ptr(ln_b), L.ptr(w1),
           L.ptr(b1), L.ptr(w2g_t), L.ptr(w1_t), L.ptr(dln_w), L.ptr(dln_b), L.dtype_code(xmid.dtype), M, C, float(eps),
           L.stream_of(xmid))
    return dxmid


def mlp_bwd_recompute_wgrad(dxout: Tensor, xmid: Tensor, ln_w: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2g_t: Tensor,
                            dw1: Tensor, db1: Tensor, s2: Tensor, cs2: Tensor, eps: float) -> None:
    """Weight-gradient half of the recompute backward: dw1, db1, s2 (raw), cs2 (raw) += ."""
    C = xmid.shape[-1]
    M = xmid.numel() // C
    ws = _mlp_ws(xmid)
    L.call('rvt_mlp_bwd_recompute_wgrad', L.ptr(dxout), L.ptr(xmid), L.ptr(ln_w), L.ptr(ln_b), L.ptr(w1), L.ptr(b1),
           L.ptr(w2g_t), L.ptr(dw1), L.ptr(db1), L.ptr(s2), L.ptr(cs2), L.ptr(ws), L.dtype_code(xmid.dtype), M, C, float(eps),
           L.stream_of(xmid))


def mlp_bwd_both_supported(dtype: torch.dtype, C: int) -> bool:
    return bool(L.get_lib().rvt_mlp_bwd_both_supported(L.dtype_code(dtype), C))


def mlp_bwd_recompute_both(dxout: Tensor, xmid: Tensor, ln_w: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2g_t: Tensor, w1_t: Tensor,
                           dln_w: Tensor, dln_b: Tensor, dw1: Tensor, db1: Tensor, s2: Tensor, cs2: Tensor, eps: float) -> Tensor:
    """The whole recompute backward of the MLP half in one launch: returns dxmid; dln_w, dln_b, dw1, db1, s2 (raw), cs2 (raw) += ."""
    C = xmid.shape[-1]
    M = xmid.numel() // C
    dxmid = torch.empty_like(xmid)
    ws = _mlp_ws(xmid)
    L.call('rvt_mlp_bwd_recompute_both', L.ptr(dxout), L.ptr(xmid), L.ptr(dxmid), L.ptr(ln_w), L.ptr(ln_b), L.ptr(w1), L.ptr(b1),
           L.ptr(w2g_t), L.ptr(w1_t), L.ptr(dln_w), L.ptr(dln_b), L.ptr(dw1), L.ptr(db1), L.ptr(s2), L.ptr(cs2), L.ptr(ws),
           L.dtype_code(xmid.dtype), M, C, float(eps), L.stream_of(xmid))
    return dxmid


def linear_dgrad(dy: Tensor, wt: Tensor, gelu_pre: Optional[Tensor] = None, add: Optional[Tensor] = None,
                 mul: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """dx = dy @ wt.T with wt = W^T stored [K][N] (optionally folded with LayerScale), then * gelu'(gelu_pre),
    + add or * mul."""
    N = dy.shape[-1]
    M = dy.numel() // N
    K = wt.shape[0]
    assert wt.shape[1] == N
    dx = _out(dy, (*dy.shape[:-1], K), out=out)
    L.call('rvt_linear_dgrad', L.ptr(dy), L.ptr(wt), L.ptr(gelu_pre), L.ptr(add), L.ptr(mul), L.ptr(dx),
           L.dtype_code(dy.dtype), M, N, K, L.stream_of(dy))
    return dx


def linear_wgrad(dy: Tensor, x: Tensor, dw: Tensor, gelu_in: bool = False, colsum_out: Optional[Tensor] = None) -> None:
    """dw += dy^T f(x); colsum_out (fp32 [N]) += column sums of dy when given (bias gradient, same pass)."""
    N = dy.shape[-1]
    K = x.shape[-1]
    M = dy.numel() // N
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (N, K) and x.numel() // K == M
    assert colsum_out is None or (colsum_out.dtype == torch.float32 and colsum_out.numel() == N)
    ws = _wgrad_ws(dy, N, K, M, colsum_out is not None)
    L.call('rvt_linear_wgrad', L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(colsum_out), L.ptr(ws), L.dtype_code(dy.dtype),
           M, N, K, int(gelu_in), L.stream_of(dy))


def attn_fwd(qkv: Tensor, F_: int, H: int, W: int, C: int, dh: int, ph: int, pw: int, window: bool,
             out: Optional[Tensor] = None) -> Tensor:
    o = _out(qkv, (F_, H, W, C), out=out)
    L.call('rvt_attn_fwd', L.ptr(qkv), L.ptr(o), L.dtype_code(qkv.dtype), F_, H, W, C, dh, ph, pw, int(window),
           L.stream_of(qkv))
    return o


def attn_bwd(qkv: Tensor, dout: Tensor, F_: int, H: int, W: int, C: int, dh: int, ph: int, pw: int, window: bool,
             out: Optional[Tensor] = None) -> Tensor:
    d = _out(qkv, qkv.shape, out=out)
    L.call('rvt_attn_bwd', L.ptr(qkv), L.ptr(dout), L.ptr(d), L.dtype_code(qkv.dtype), F_, H, W, C, dh, ph, pw,
           int(window), L.stream_of(qkv))
    return d


def attn_block_supported(dtype: torch.dtype, C: int, dh: int, n_tok: int) -> bool:
    return dtype in L._DT and bool(L.get_lib().rvt_attn_block_supported(L.dtype_code(dtype), C, dh, n_tok))


def attn_block_fwd(x: Tensor, ln_w: Optional[Tensor], ln_b: Optional[Tensor], wqkv: Tensor, bqkv: Tensor, wp: Tensor, bp: Tensor,
                   gamma: Tensor, F_: int, H: int, W: int, C: int, dh: int, ph: int, pw: int, window: bool, eps: float,
                   want_a: bool):
    """Fused attention half (csrc/attn_block.hpp): xmid = x + gamma * (attention(LN1(x) wqkv^T + bqkv) wp^T + bp).
    Returns (xmid, a) with a = the attention output rows (kept for the proj weight gradient) or None."""
    xmid = torch.empty_like(x)
    a = torch.empty_like(x) if want_a else None
    L.call('rvt_attn_block_fwd', L.ptr(x), L.ptr(xmid), L.ptr(a), L.ptr(ln_w), L.ptr(ln_b), L.ptr(wqkv), L.ptr(bqkv), L.ptr(wp),
           L.ptr(bp), L.ptr(gamma), L.dtype_code(x.dtype), F_, H, W, C, dh, ph, pw, int(window), float(eps), L.stream_of(x))
    return xmid, a


def attn_block_bwd(x: Tensor, dxmid: Tensor, ln_w: Optional[Tensor], ln_b: Optional[Tensor], wqkv: Tensor, bqkv: Tensor,
                   wpg_t: Tensor, dln_w: Optional[Tensor], dln_b: Optional[Tensor], F_: int, H: int, W: int, C: int, dh: int,
                   ph: int, pw: int, window: bool, eps: float):
    """Backward of the fused attention half from (x, dxmid): returns (dx, dqkv, u) with u = LN1(x) (None without norm1:
    then x itself is the operand of the qkv weight gradient); dln_w / dln_b are accumulated."""
    dx = torch.empty_like(x)
    dqkv = torch.empty((*x.shape[:-1], 3 * C), dtype=x.dtype, device=x.device)
    u = torch.empty_like(x) if ln_w is not None else None
    L.call('rvt_attn_block_bwd', L.ptr(x), L.ptr(dxmid), L.ptr(dx), L.ptr(dqkv), L.ptr(u), L.ptr(ln_w), L.ptr(ln_b), L.ptr(wqkv),
           L.ptr(bqkv), L.ptr(wpg_t), L.ptr(dln_w), L.ptr(dln_b), L.dtype_code(x.dtype), F_, H, W, C, dh, ph, pw, int(window),
           float(eps), L.stream_of(x))
    return dx, dqkv, u


def attn_block_bwd_preln(x: Tensor, y0: Tensor, dxmid: Tensor, ln_w: Tensor, wqkv: Tensor, bqkv: Tensor, wpg_t: Tensor, dln_w: Tensor,
                         dln_b: Tensor, F_: int, H: int, W: int, C: int, dh: int, ph: int, pw: int, window: bool, eps: float):
    """Backward of the fused attention half of a stage's FIRST block (no norm1) carried through the LayerNorm in front of it,
    x = LN(y0) (the down-sampling norm, maxvit.py:177): returns (dy0, dqkv); dln_w / dln_b of THAT norm are accumulated.
    Same dqkv as attn_block_bwd(ln_w=None); dy0 = layernorm_bwd(y0, ln_w, dx) of its dx, in one launch."""
    dy0 = torch.empty_like(x)
    dqkv = torch.empty((*x.shape[:-1], 3 * C), dtype=x.dtype, device=x.device)
    L.call('rvt_attn_block_bwd_preln', L.ptr(x), L.ptr(y0), L.ptr(dxmid), L.ptr(dy0), L.ptr(dqkv), L.ptr(ln_w), L.ptr(wqkv), L.ptr(bqkv),
           L.ptr(wpg_t), L.ptr(dln_w), L.ptr(dln_b), L.dtype_code(x.dtype), F_, H, W, C, dh, ph, pw, int(window), float(eps),
           L.stream_of(x))
    return dy0, dqkv


def lstm_fwd(x: Tensor, h_prev: Tensor, c_prev: Tensor, w_perm: Tensor, b_perm: Tensor, h_out: Tensor, c_out: Tensor,
             gates: Optional[Tensor]) -> None:
    C = x.shape[-1]
    M = x.numel() // C
    assert c_prev.dtype == torch.float32 and c_out.dtype == torch.float32
    L.call('rvt_lstm_fwd', L.ptr(x), L.ptr(h_prev), L.ptr(c_prev), L.ptr(w_perm), L.ptr(b_perm), L.ptr(h_out),
           L.ptr(c_out), L.ptr(gates), L.dtype_code(x.dtype), M, C, L.stream_of(x))


def lstm_gates_bwd(dh_in: Tensor, dh_rec: Optional[Tensor], dc_rec: Tensor, gates: Tensor, c_new: Tensor,
                   c_prev: Tensor, dz: Tensor) -> None:
    C = dh_in.shape[-1]
    M = dh_in.numel() // C
    L.call('rvt_lstm_gates_bwd', L.ptr(dh_in), L.ptr(dh_rec), L.ptr(dc_rec), L.ptr(gates), L.ptr(c_new), L.ptr(c_prev),
           L.ptr(dz), L.dtype_code(dh_in.dtype), M, C, L.stream_of(dh_in))


def lstm_dgrad(dz: Tensor, wt: Tensor, dx: Tensor, dh_rec: Tensor) -> None:
    C = dx.shape[-1]
    M = dx.numel() // C
    L.call('rvt_lstm_dgrad', L.ptr(dz), L.ptr(wt), L.ptr(dx), L.ptr(dh_rec), L.dtype_code(dz.dtype), M, C,
           L.stream_of(dz))


def lstm_wgrad(dz: Tensor, x: Tensor, h_prev: Tensor, dw: Tensor, colsum_out: Optional[Tensor] = None) -> None:
    C = x.shape[-1]
    M = x.numel() // C
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (4 * C, 2 * C)
    ws = _wgrad_ws(dz, 4 * C, 2 * C, M, colsum_out is not None)
    L.call('rvt_lstm_wgrad', L.ptr(dz), L.ptr(x), L.ptr(h_prev), L.ptr(dw), L.ptr(colsum_out), L.ptr(ws),
           L.dtype_code(dz.dtype), M, C, L.stream_of(dz))


def lstm_scan_supported(dtype: torch.dtype, C: int) -> bool:
    return bool(L.get_lib().rvt_lstm_scan_supported(L.dtype_code(dtype), C))


def lstm_scan_saves_gates(dtype: torch.dtype, C: int) -> bool:
    """bf16 C = 128: the scan keeps its weights in the register file and saves the activated gates for the reverse scan."""
    return dtype in L._DT and bool(L.get_lib().rvt_lstm_scan_saves_gates(L.dtype_code(dtype), C))


def lstm_scan_fwd(x_all: Tensor, Hall: Tensor, c0: Optional[Tensor], c_last: Tensor, Csave: Optional[Tensor], w: Tensor,
                  bias: Tensor, gates_out: Optional[Tensor] = None) -> None:
    """All T steps of the 1x1-conv ConvLSTM in one launch.  x_all (T,M..,C); Hall (T+1,M..,C) with slot 0 = incoming h
    (filled by the caller); c0 fp32 or None (zeros); c_last fp32 out; Csave (T,M..,C) copy of the cell states for BPTT or
    None; w [4C][2C] natural gate order; gates_out (T,M..,4C) or None (only where lstm_scan_saves_gates)."""
    T_, C = x_all.shape[0], x_all.shape[-1]
    M = x_all[0].numel() // C
    assert c_last.dtype == torch.float32 and (c0 is None or c0.dtype == torch.float32)
    L.call('rvt_lstm_scan_fwd', L.ptr(x_all), L.ptr(Hall), L.ptr(c0), L.ptr(c_last), L.ptr(Csave), L.ptr(w), L.ptr(bias),
           L.ptr(gates_out), L.dtype_code(x_all.dtype), M, C, T_, L.stream_of(x_all))


def lstm_scan_wgrad_supported(dtype: torch.dtype, C: int, M: int) -> bool:
    """In-kernel weight gradients of the reverse scan (bf16, LDS-resident weights: C <= 64)."""
    return dtype in L._DT and L.get_lib().rvt_lstm_scan_bwd_ws_floats(L.dtype_code(dtype), C, M) > 0


def lstm_scan_bwd(x_all: Tensor, Hall: Tensor, Csave: Tensor, c0: Optional[Tensor], dH: Optional[Tensor],
                  dc_last: Optional[Tensor], w: Tensor, wt: Tensor, bias: Tensor, dx_all: Tensor, dz_all: Optional[Tensor],
                  dh0: Tensor, dc0: Tensor, dw: Optional[Tensor] = None, db: Optional[Tensor] = None,
                  gates: Optional[Tensor] = None) -> None:
    """Reverse scan.  With dw / db (fp32 [4C][2C] / [4C], +=) the weight gradients are accumulated inside the kernel and
    dz_all is not produced (pass None).  With `gates` (saved by lstm_scan_fwd where lstm_scan_saves_gates) the gates are read
    instead of recomputed."""
    T_, C = x_all.shape[0], x_all.shape[-1]
    M = x_all[0].numel() // C
    assert dc0.dtype == torch.float32 and (dc_last is None or dc_last.dtype == torch.float32)
    ws = None
    st = L.stream_of(x_all)
    if dw is not None:
        assert dw.dtype == torch.float32 and db is not None and db.dtype == torch.float32 and tuple(dw.shape) == (4 * C, 2 * C)
        n = L.get_lib().rvt_lstm_scan_bwd_ws_floats(L.dtype_code(x_all.dtype), C, M)
        key = ('scanbwd', x_all.device.type, x_all.device.index, 0 if st is None else int(st))
        ws = _WS.get(key)
        if ws is None or ws.numel() < n:
            ws = torch.empty(n, dtype=torch.float32, device=x_all.device)
            _WS[key] = ws
    L.call('rvt_lstm_scan_bwd', L.ptr(x_all), L.ptr(Hall), L.ptr(Csave), L.ptr(c0), L.ptr(dH), L.ptr(dc_last), L.ptr(w),
           L.ptr(wt), L.ptr(bias), L.ptr(dx_all), L.ptr(dz_all), L.ptr(dh0), L.ptr(dc0), L.ptr(dw), L.ptr(db), L.ptr(ws),
           L.ptr(gates), L.dtype_code(x_all.dtype), M, C, T_, st)


def lstm_scan3_supported(dtype: torch.dtype, C: int) -> bool:
    """ConvLSTM of the wide stages with the time loop in the kernel and streamed weights (csrc/lstm_scan3.hpp): bf16, C = 256."""
    return dtype in L._DT and bool(L.get_lib().rvt_lstm_scan3_supported(L.dtype_code(dtype), C))


def lstm_scan3_rows(C: int, M: int) -> int:
    """Rows per time step of the register-dump buffers (gates, cell states) the forward saves for the reverse scan."""
    return int(L.get_lib().rvt_lstm_scan3_rows(C, M))


def lstm_scan3_pack(w: Tensor, fwd: bool = True, bwd: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """Natural [4C][2C] ConvLSTM weights -> the operand-order streams of lstm_scan3_fwd / lstm_scan3_bwd."""
    C = w.shape[1] // 2
    assert tuple(w.shape) == (4 * C, 2 * C) and w.is_contiguous()
    wp = torch.empty(4 * C * 2 * C, dtype=w.dtype, device=w.device) if fwd else None
    wtp = torch.empty(4 * C * 2 * C, dtype=w.dtype, device=w.device) if bwd else None
    L.call('rvt_lstm_scan3_pack', L.ptr(w), L.ptr(wp), L.ptr(wtp), C, L.stream_of(w))
    return wp, wtp


def lstm_scan3_fwd(x_all: Tensor, Hall: Tensor, c0: Optional[Tensor], c_last: Tensor, Csave: Optional[Tensor], wp: Tensor,
                   bias: Tensor, gsave: Optional[Tensor]) -> None:
    """All T steps in one launch; Csave / gsave: dump buffers of T * lstm_scan3_rows(C, M) * C / * 4C elements, or both None."""
    T_, C = x_all.shape[0], x_all.shape[-1]
    M = x_all[0].numel() // C
    assert c_last.dtype == torch.float32 and (c0 is None or c0.dtype == torch.float32) and bias.dtype == torch.float32
    if gsave is not None:
        rows = lstm_scan3_rows(C, M)
        assert gsave.numel() >= T_ * rows * 4 * C and Csave.numel() >= T_ * rows * C
    L.call('rvt_lstm_scan3_fwd', L.ptr(x_all), L.ptr(Hall), L.ptr(c0), L.ptr(c_last), L.ptr(Csave), L.ptr(wp), L.ptr(bias),
           L.ptr(gsave), L.dtype_code(x_all.dtype), M, C, T_, L.stream_of(x_all))


def lstm_scan3_bwd(gsave: Tensor, Csave: Tensor, c0: Optional[Tensor], dH: Optional[Tensor], dc_last: Optional[Tensor], wtp: Tensor,
                   dx_all: Tensor, dz_all: Tensor, dh0: Tensor, dc0: Tensor) -> None:
    T_, C = dx_all.shape[0], dx_all.shape[-1]
    M = dx_all[0].numel() // C
    assert dc0.dtype == torch.float32 and (dc_last is None or dc_last.dtype == torch.float32)
    L.call('rvt_lstm_scan3_bwd', L.ptr(gsave), L.ptr(Csave), L.ptr(c0), L.ptr(dH), L.ptr(dc_last), L.ptr(wtp), L.ptr(dx_all),
           L.ptr(dz_all), L.ptr(dh0), L.ptr(dc0), L.dtype_code(dx_all.dtype), M, C, T_, L.stream_of(dx_all))


def dwconv(x: Tensor, w: Tensor, b: Optional[Tensor], k: int, transpose: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """Depth-wise k x k conv on (N,H,W,C) channels-last (rnn.py:25-29); transpose=True -> input gradient."""
    N, H, W, C = x.shape
    assert w.dtype == torch.float32 and tuple(w.shape) == (C, k * k)
    y = _out(x, x.shape, out=out)
    L.call('rvt_dwconv_fwd', L.ptr(x), C, L.ptr(w), L.ptr(b), L.ptr(y), C, L.dtype_code(x.dtype), N, H, W, C, k,
           int(transpose), L.stream_of(x))
    return y


def dwconv_wgrad(x: Tensor, dy: Tensor, dw: Tensor, db: Tensor, k: int) -> None:
    N, H, W, C = x.shape
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (C, k * k) and db.numel() == C
    L.call('rvt_dwconv_wgrad', L.ptr(x), C, L.ptr(dy), C, L.ptr(dw), L.ptr(db), L.dtype_code(x.dtype), N, H, W, C, k,
           L.stream_of(x))


def token_mask_fwd(x: Tensor, mask_u8: Tensor, token: Tensor) -> None:
    """x[mask] = token in place (maxvit_rnn.py:174-176); x (…,C), mask_u8 uint8 over the leading dims, token fp32 [C]."""
    C = x.shape[-1]
    L.call('rvt_token_mask_fwd', L.ptr(x), L.ptr(mask_u8), L.ptr(token), L.dtype_code(x.dtype), x.numel() // C, C,
           L.stream_of(x))


def token_mask_bwd(dx: Tensor, mask_u8: Tensor, dtoken: Tensor) -> None:
    C = dx.shape[-1]
    L.call('rvt_token_mask_bwd', L.ptr(dx), L.ptr(mask_u8), L.ptr(dtoken), L.dtype_code(dx.dtype), dx.numel() // C, C,
           L.stream_of(dx))


def state_reset_masked(st: Tensor, mask: Tensor) -> None:
    """Zero rows st[b] where mask[b] (modules/utils/detection.py:96-113); st is (B, ...)."""
    B = st.shape[0]
    m = mask.to(device=st.device, dtype=torch.uint8).contiguous()
    L.call('rvt_state_reset_masked', L.ptr(st), L.ptr(m), L.dtype_code(st.dtype), B, st.numel() // B, L.stream_of(st))


class _GatherFrames(torch.autograd.Function):
    """frames (N, ...) contiguous -> frames[idx] (one HIP gather; backward = one scatter into zeros)."""

    @staticmethod
    def forward(ctx, frames: Tensor, idx: Tensor) -> Tensor:
        n = int(idx.numel())
        out = torch.empty((n, *frames.shape[1:]), dtype=frames.dtype, device=frames.device)
        fb = frames[0].numel() * frames.element_size()
        L.call('rvt_gather_frames', L.ptr(frames), L.ptr(idx), L.ptr(out), n, fb, 0, L.stream_of(frames))
        ctx.save_for_backward(idx)
        ctx.shape = tuple(frames.shape)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        d = torch.zeros(ctx.shape, dtype=dout.dtype, device=dout.device)
        fb = d[0].numel() * d.element_size()
        L.call('rvt_gather_frames', L.ptr(dout), L.ptr(idx), L.ptr(d), int(idx.numel()), fb, 1, L.stream_of(dout))
        return d, None


def gather_frames(frames: Tensor, idx: Tensor) -> Tensor:
    """frames: (N, ...) contiguous; idx: int32 device tensor of DISTINCT frame indices.  Differentiable in `frames`."""
    assert idx.dtype == torch.int32 and frames.is_contiguous()
    return _GatherFrames.apply(frames, idx)

