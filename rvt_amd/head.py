"""YOLOX detection head, SimOTA label assignment and the detection losses — MI355X-native.

Mirror of the reference module surface (models/detection/yolox/models/yolo_head.py:20-606, built by
models/detection/yolox_extension/models/build.py:9-18): same constructor arguments, same parameter / buffer names and shapes
(`stems.0.conv.weight`, `cls_convs.1.0.bn.running_mean`, `reg_preds.2.bias`, ...: reference checkpoints load with strict=True),
same `forward(xin, labels=None) -> (outputs, losses)` with `outputs` = decoded detections [B][A][5 + num_classes] and, in
training mode, `losses` = {"loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg"}.

What runs where:
  * stems / class tower / regression tower: BaseConv units (rvt_amd/fpn.py: conv on the GEMM engine + BatchNorm + SiLU kernels);
  * the three 1x1 prediction convolutions per level (yolo_head.py:107-133): two GEMMs per level on the linear engine —
    [reg(4) | obj(1)] share the regression tower's output and run as ONE N = 8 GEMM (rows zero-padded to the engine's 8-column
    granularity), the class predictor as another; rvt_linear_fwd / _dgrad / _wgrad with the bias fused;
  * decode (:248-290), SimOTA (:453-606) and the losses (:291-443): rvt_yolox_decode / rvt_simota_loss / rvt_yolox_decode_bwd
    (csrc/simota.hpp) — batched over the images, no per-image Python loop, no host synchronisation anywhere in the step tail
    (the reference has int(nlabel[b]) :325, .item() :596, a per-ground-truth topk loop :580-584 and empty_cache :383).

Differences a caller can see: `losses["num_fg"]` is a 0-dim device tensor instead of a Python float (reading it is the caller's
choice of sync point); `outputs` carries no autograd history (the reference returns it attached, nothing in the reference
differentiates it: modules/detection.py uses it for the detections only); use_l1 (never switched on by RVT), depthwise and
decode_in_inference=False are not built; an image whose ground truths have no candidate anchor at all gets no matches where the
reference's torch.topk raises.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .fpn import BaseConv, ConvPack

Tensor = torch.Tensor


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class _PredFn(torch.autograd.Function):
    """y = x @ W^T + b on a channels-last map, W [N][K] with N a multiple of 8 (the padded 1x1 prediction convolutions)."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Tensor):
        dt = x.dtype
        wk = w.detach().to(dt).contiguous()
        y = ops.linear_fwd(x, wk, b.detach().float().contiguous())
        ctx.save_for_backward(x, wk)
        ctx.wdtype = w.dtype
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, wk = ctx.saved_tensors
        dy = dy.contiguous()
        N, K = wk.shape
        dx = ops.linear_dgrad(dy, wk.t().contiguous()) if ctx.needs_input_grad[0] else None
        dw = torch.zeros(N, K, dtype=torch.float32, device=x.device)
        db = torch.zeros(N, dtype=torch.float32, device=x.device)
        ops.linear_wgrad(dy, x, dw, colsum_out=db)
        return dx, dw.to(ctx.wdtype), db.to(ctx.wdtype)


class _Levels:
    """Host-side level table handed to the C entry points (anchors are ordered level by level, yolo_head.py:236-241)."""

    def __init__(self, hws: Sequence[Tuple[int, int]], strides: Sequence[int]):
        self.hws, self.strides = [tuple(h) for h in hws], [int(s) for s in strides]
        self.L = len(hws)
        self.hw_arr = (ctypes.c_int * (2 * self.L))(*[v for hw in self.hws for v in hw])
        self.st_arr = (ctypes.c_int * self.L)(*self.strides)
        self.a0 = [0]
        for h, w in self.hws:
            self.a0.append(self.a0[-1] + h * w)
        self.A = self.a0[-1]


def _decode_levels(lv: _Levels, maps: Sequence[Tensor], B: int, nc: int, pred_train: Optional[Tensor], pred_infer: Optional[Tensor]) -> None:
    for l in range(lv.L):
        ro, cl = maps[2 * l], maps[2 * l + 1]
        H, W = lv.hws[l]
        L.call('rvt_yolox_decode', L.ptr(ro), L.ptr(cl), ro.shape[-1], cl.shape[-1], L.dtype_code(ro.dtype), B, H, W, lv.strides[l], nc,
               lv.a0[l], lv.A, L.ptr(pred_train), L.ptr(pred_infer), L.stream_of(ro))


def decode(maps: Sequence[Tensor], hws, strides, num_classes: int) -> Tensor:
    """Inference tail (yolo_head.py:211-246, :269-290): per-level [reg|obj] and class maps -> detections [B][A][5+nc] fp32."""
    lv = _Levels(hws, strides)
    B = maps[0].shape[0]
    out = torch.empty(B, lv.A, 5 + num_classes, dtype=torch.float32, device=maps[0].device)
    _decode_levels(lv, [m.contiguous() for m in maps], B, num_classes, None, out)
    return out


class _HeadLossFn(torch.autograd.Function):
    """(labels, per-level prediction maps) -> (detections, losses[5]); the gradient of every loss component flows to the maps."""

    @staticmethod
    def forward(ctx, labels: Tensor, lv: _Levels, nc: int, *maps: Tensor):
        maps = tuple(m.contiguous() for m in maps)
        dev = maps[0].device
        B, G = maps[0].shape[0], labels.shape[1]
        NO, A = 5 + nc, lv.A
        f32 = torch.float32
        lab = labels.detach().to(device=dev, dtype=f32).contiguous()
        pred_train = torch.empty(B, A, NO, dtype=f32, device=dev)
        pred_infer = torch.empty(B, A, NO, dtype=f32, device=dev)
        _decode_levels(lv, maps, B, nc, pred_train, pred_infer)
        need_grad = any(ctx.needs_input_grad[3:])
        g_pred = torch.empty_like(pred_train) if need_grad else None
        losses = torch.empty(5, dtype=f32, device=dev)
        match = torch.empty(B, A, dtype=torch.int32, device=dev)
        piou = torch.empty(B, A, dtype=f32, device=dev)
        ws_bytes = L.get_lib().rvt_simota_ws_bytes(B, G, A)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.call('rvt_simota_loss', L.ptr(pred_train), L.ptr(lab), lv.hw_arr, lv.st_arr, lv.L, B, G, A, nc, L.ptr(losses), L.ptr(g_pred),
               L.ptr(match), L.ptr(piou), L.ptr(ws), ws_bytes, L.stream_of(pred_train))
        ctx.lv, ctx.nc, ctx.B = lv, nc, B
        ctx.shapes = [(m.shape, m.dtype) for m in maps]
        if need_grad:
            ctx.save_for_backward(g_pred, pred_train)
        ctx.mark_non_differentiable(pred_infer, match, piou)
        return pred_infer, losses, match, piou

    @staticmethod
    def backward(ctx, _g_out, g_losses, _g_match, _g_piou):
        g_pred, pred_train = ctx.saved_tensors
        lv, nc, B = ctx.lv, ctx.nc, ctx.B
        g = g_losses.float()
        # losses = (loss, 5 iou, obj, cls, num_fg ratio), loss = 5 iou + obj + cls; g_pred holds d iou / d obj / d cls column-wise
        col_scale = torch.stack([5.0 * (g[0] + g[1]), g[0] + g[2], g[0] + g[3]]).contiguous()
        grads = []
        for l in range(lv.L):
            (s_ro, dt), (s_cl, _) = ctx.shapes[2 * l], ctx.shapes[2 * l + 1]
            d_ro = torch.empty(s_ro, dtype=dt, device=g_pred.device)
            d_cl = torch.empty(s_cl, dtype=dt, device=g_pred.device)
            H, W = lv.hws[l]
            L.call('rvt_yolox_decode_bwd', L.ptr(g_pred), L.ptr(pred_train), L.ptr(col_scale), L.ptr(d_ro), L.ptr(d_cl), s_ro[-1], s_cl[-1],
                   L.dtype_code(dt), B, H, W, lv.strides[l], nc, lv.a0[l], lv.A, L.stream_of(g_pred))
            grads += [d_ro, d_cl]
        return (None, None, None, *grads)


def simota_loss(maps: Sequence[Tensor], labels: Tensor, hws, strides, num_classes: int):
    """Functional form of the training tail: returns (detections, losses[5], match [B][A] int32, matched IoU [B][A])."""
    return _HeadLossFn.apply(labels, _Levels(hws, strides), num_classes, *maps)


class YOLOXHead(nn.Module):
    def __init__(self, num_classes: int = 80, strides: Tuple[int, ...] = (8, 16, 32), in_channels: Tuple[int, ...] = (256, 512, 1024),
                 act: str = 'silu', depthwise: bool = False, compile_cfg: Optional[Dict] = None,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        if depthwise:
            raise NotImplementedError('depthwise YOLOX head is not built (no shipped config enables it)')
        self.num_classes, self.strides, self.compute_dtype = num_classes, tuple(int(s) for s in strides), compute_dtype
        self.decode_in_inference = True
        self.use_l1 = False
        hidden = int(256 * (in_channels[-1] / 1024))                      # yolo_head.py:46-54 (width scaling from the last stage)
        if hidden % 8:
            raise NotImplementedError(f'head width {hidden} must be a multiple of 8')
        self.hidden_dim = hidden
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()          # (registration order = the reference's state_dict order)
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        for c in in_channels:
            self.stems.append(BaseConv(int(c), hidden, 1, 1, act=act))
            self.cls_convs.append(nn.Sequential(BaseConv(hidden, hidden, 3, 1, act=act), BaseConv(hidden, hidden, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(BaseConv(hidden, hidden, 3, 1, act=act), BaseConv(hidden, hidden, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(hidden, num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hidden, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hidden, 1, 1, 1, 0))
        self.initialize_biases(prior_prob=0.01)
        self._pack = ConvPack(self)
        self._pred_cache: Dict[int, tuple] = {}

    def initialize_biases(self, prior_prob: float) -> None:
        """Focal-loss prior on the class and objectness biases (yolo_head.py:155-165)."""
        v = -math.log((1 - prior_prob) / prior_prob)
        for conv in list(self.cls_preds) + list(self.obj_preds):
            with torch.no_grad():
                conv.bias.fill_(v)

    def _pred_maps(self, xin: Sequence[Tensor]):
        dt, nc, hid = self.compute_dtype, self.num_classes, self.hidden_dim
        self._pack.refresh(dt, self.training)
        maps, hws = [], []
        for k, x in enumerate(xin):
            x = x.permute(0, 2, 3, 1).to(dt).contiguous()                 # channels-last (free for the FPN's own outputs)
            hws.append((x.shape[1], x.shape[2]))
            x = self.stems[k](x)
            cls_feat = self.cls_convs[k](x)
            reg_feat = self.reg_convs[k](x)
            # the three 1x1 convolutions as two 8-row-aligned GEMMs; the padding rows are constants (zero weight, zero bias)
            infer = not self.training and not torch.is_grad_enabled()
            w_ro, b_ro, w_cl, b_cl = self._pred_weights(k, x, cache=infer)
            if infer:                                                      # no autograd node, weights already in the compute dtype
                maps += [ops.linear_fwd(reg_feat, w_ro, b_ro), ops.linear_fwd(cls_feat, w_cl, b_cl)]
            else:
                maps += [_PredFn.apply(reg_feat, w_ro, b_ro), _PredFn.apply(cls_feat, w_cl, b_cl)]
        return maps, hws

    def _pred_weights(self, k: int, like: Tensor, cache: bool):
        """[reg(4) | obj(1) | 0 0 0] and [cls(nc) | 0 ..] weight / bias blocks of level k.  Training: built per call (autograd reaches the
        three Conv2d parameters through the concatenation); inference: cast to the compute dtype and cached per parameter version."""
        nc, hid = self.num_classes, self.hidden_dim
        ps = (self.reg_preds[k].weight, self.reg_preds[k].bias, self.obj_preds[k].weight, self.obj_preds[k].bias,
              self.cls_preds[k].weight, self.cls_preds[k].bias)
        key = (like.dtype, like.device, tuple((p.data_ptr(), p._version) for p in ps))
        if cache:
            hit = self._pred_cache.get(k)
            if hit is not None and hit[0] == key:
                return hit[1]
        np_ = _pad8(nc)
        w_ro = torch.cat([ps[0].reshape(4, hid), ps[2].reshape(1, hid), like.new_zeros(3, hid, dtype=ps[0].dtype)])
        b_ro = torch.cat([ps[1], ps[3], like.new_zeros(3, dtype=ps[1].dtype)])
        w_cl = torch.cat([ps[4].reshape(nc, hid), like.new_zeros(np_ - nc, hid, dtype=ps[4].dtype)])
        b_cl = torch.cat([ps[5], like.new_zeros(np_ - nc, dtype=ps[5].dtype)])
        out = (w_ro, b_ro, w_cl, b_cl)
        if cache:
            out = (w_ro.detach().to(like.dtype).contiguous(), b_ro.detach().float().contiguous(),
                   w_cl.detach().to(like.dtype).contiguous(), b_cl.detach().float().contiguous())
            self._pred_cache[k] = (key, out)
        return out

    def forward(self, xin: Sequence[Tensor], labels: Optional[Tensor] = None):
        """xin: the FPN maps, (N, C, H, W)-shaped; labels (training): [B][G][5] rows (class, cx, cy, w, h), zero rows pad."""
        if not self.decode_in_inference:
            raise NotImplementedError('decode_in_inference=False (an export option) is not built')
        maps, hws = self._pred_maps(xin)
        self.hw = hws
        if not self.training:
            return decode(maps, hws, self.strides, self.num_classes), None
        assert labels is not None, 'training mode needs labels'
        outputs, ls, _match, _piou = _HeadLossFn.apply(labels, _Levels(hws, self.strides), self.num_classes, *maps)
        self.last_match, self.last_matched_iou = _match, _piou              # diagnostics (device tensors; nothing is synchronised)
        losses = {'loss': ls[0], 'iou_loss': ls[1], 'conf_loss': ls[2], 'cls_loss': ls[3], 'l1_loss': 0.0, 'num_fg': ls[4].detach()}
        return outputs, losses


def build_yolox_head(head_cfg, in_channels: Tuple[int, ...], strides: Tuple[int, ...], compute_dtype: torch.dtype = torch.float32) -> YOLOXHead:
    """Registry entry point (reference yolox_extension/models/build.py:9-18)."""
    d = dict(head_cfg)
    d.pop('name', None)
    d.pop('version', None)
    d.pop('compile', None)
    return YOLOXHead(in_channels=tuple(in_channels), strides=tuple(strides), compute_dtype=compute_dtype, **d)
