"""YOLOX PAFPN on the backbone's stage 2-4 features — MI355X-native.

Mirror of the reference module surface (models/detection/yolox_extension/models/yolo_pafpn.py:18-139 built from the blocks of
models/detection/yolox/models/network_blocks.py:29-141): same constructor arguments, same parameter / buffer names and shapes
(`lateral_conv0.conv.weight`, `C3_p4.m.0.conv1.bn.running_var`, ...: reference checkpoints load with strict=True), same
`forward(input: Dict[int, Tensor]) -> (pan_out2, pan_out1, pan_out0)`.

Every BaseConv unit (Conv2d(bias=False) -> BatchNorm2d -> SiLU) runs on the HIP kernels behind include/rvt_hip.h, forward AND
backward, channels-last end to end: the convolution on the im2col GEMM engine (rvt_conv_fwd / _dgrad / _wgrad), BatchNorm
(batch statistics in training, running statistics in eval, running-statistics update) + SiLU on the row-wise kernels of
csrc/bnact.hpp.  torch is used for memory and for the two pure data movements of the neck (channel concatenation and nearest
2x up-sampling of channels-last maps), whose autograd is index arithmetic.  Under data parallelism the two statistics vectors of
each BatchNorm are all-reduced (what SyncBatchNorm does, train.py:133).

Not built: depthwise=True (DWConv, no shipped config), activations other than SiLU.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import ops, weights

Tensor = torch.Tensor
BN_ACT_SILU = 1


def _bn_reduce(t: Tensor) -> None:
    """SyncBatchNorm's exchange: sum the statistics over the data-parallel ranks (no-op at world size 1)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)


class _BaseConvFn(torch.autograd.Function):
    """y = SiLU(BatchNorm(conv(x))) on a channels-last (N, H, W, Cin) map; every arithmetic step is a HIP kernel."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, gamma: Tensor, beta: Tensor, mod: 'BaseConv', training: bool, sync: bool):
        dt, dev = x.dtype, x.device
        N, H, W, Cin = x.shape
        Cout, k, s, pad = mod.out_channels, mod.ksize, mod.stride, mod.pad
        pk = mod._pk if mod._pk is not None and mod._pk.dtype == dt and mod._pk.cin == Cin else None
        if not training and pk is not None and pk.fin is not None and not any(ctx.needs_input_grad[:4]):
            # inference: BatchNorm (running statistics) + SiLU in the convolution's epilogue — one launch per unit
            return ops.conv_bn_act_fwd(x, pk.wp, pk.fin[2], pk.fin[3], k, s, pad, BN_ACT_SILU)
        wp = pk.wp if pk is not None else weights.pack_conv_fwd(w.detach(), Cin, dt)      # [Cout][k*k*Cin] tap-major
        y0 = ops.conv_fwd(x, wp, k, s, pad)                                               # network_blocks.py:37-45 (bias=False)
        rows = y0.numel() // Cout
        f32 = torch.float32
        st = L.stream_of(y0)
        stats = pk.stats if pk is not None and training else torch.zeros(2, Cout, dtype=f32, device=dev)   # (the pack zeroes its arena once per forward)
        ctx.pk = pk
        count = rows
        if training:
            L.call('rvt_bn_stats', L.ptr(y0), L.ptr(stats[0]), L.ptr(stats[1]), L.dtype_code(dt), rows, Cout, st)
            if sync:
                # ONE exchange of [sum x | sum x^2 | rows] (the 2 C + 1 values torch's SyncBatchNorm gathers); the global row count comes
                # back to the host because the finalize kernel takes it by value (ranks may hold different numbers of labelled frames)
                packed = torch.cat([stats.reshape(-1), torch.tensor([float(rows)], device=dev)])
                _bn_reduce(packed)
                stats.copy_(packed[:2 * Cout].view(2, Cout))
                count = int(packed[2 * Cout].item())
        fin = torch.empty(4, Cout, dtype=f32, device=dev)                                 # mean, rstd, scale, shift
        bn = mod.bn
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = torch.empty_like(y0)
        if not bn.track_running_stats or (training and bn.momentum is None):
            # (an eval / no-grad forward never updates the running statistics: momentum is irrelevant there)
            raise NotImplementedError('rvt_amd.fpn.BaseConv: BatchNorm2d with track_running_stats=False, or a training forward with '
                                      'momentum=None (cumulative average), is not built (no shipped config uses either)')
        mom = 0.0 if bn.momentum is None else float(bn.momentum)
        rm, rv = L.ptr(bn.running_mean), L.ptr(bn.running_var)
        if training:                                                                      # finalize + activation in one launch
            L.call('rvt_bn_train_act_fwd', L.ptr(y0), L.ptr(stats[0]), L.ptr(stats[1]), count, L.ptr(g32), L.ptr(b32), float(bn.eps), mom,
                   rm, rv, L.ptr(fin[0]), L.ptr(fin[1]), L.ptr(fin[2]), L.ptr(fin[3]), L.ptr(y), L.dtype_code(dt), rows, Cout,
                   BN_ACT_SILU, st)
            if bn.track_running_stats and pk is None:                                     # (with a ConvPack: one batched increment per forward)
                bn.num_batches_tracked += 1
        else:
            L.call('rvt_bn_finalize', L.ptr(stats[0]), L.ptr(stats[1]), count, L.ptr(g32), L.ptr(b32), float(bn.eps), mom, rm, rv,
                   L.ptr(fin[0]), L.ptr(fin[1]), L.ptr(fin[2]), L.ptr(fin[3]), Cout, 0, st)
            L.call('rvt_bn_act_fwd', L.ptr(y0), L.ptr(fin[2]), L.ptr(fin[3]), L.ptr(y), L.dtype_code(dt), rows, Cout, BN_ACT_SILU, st)
        ctx.save_for_backward(x, w, y0, fin)
        ctx.mod, ctx.count, ctx.sync, ctx.training = mod, count, sync, training
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, w, y0, fin = ctx.saved_tensors
        mod = ctx.mod
        assert ctx.training, 'the PAFPN backward is the training-mode (batch-statistics) BatchNorm backward'
        dt, dev = x.dtype, x.device
        N, H, W, Cin = x.shape
        Cout, k, s, pad = mod.out_channels, mod.ksize, mod.stride, mod.pad
        rows = y0.numel() // Cout
        dy = dy.contiguous()
        st = L.stream_of(dy)
        ds = torch.zeros(2, Cout, dtype=torch.float32, device=dev)                        # sum dz (= dbeta), sum dz * xhat (= dgamma)
        L.call('rvt_bn_act_bwd_stats', L.ptr(dy), L.ptr(y0), L.ptr(fin[2]), L.ptr(fin[3]), L.ptr(fin[0]), L.ptr(fin[1]), L.ptr(ds[0]),
               L.ptr(ds[1]), L.dtype_code(dt), rows, Cout, BN_ACT_SILU, st)
        ds_local = ds
        if ctx.sync:                                   # the input gradient needs the GLOBAL sums; dgamma / dbeta stay this rank's share
            ds = ds.clone()                            # (the data-parallel gradient reduction sums / averages them like every other parameter)
            _bn_reduce(ds)
        dconv = torch.empty_like(y0)
        # (the kernel divides the two sums by its `rows` argument; with synchronised statistics the divisor is the global count)
        dsk = ds if ctx.count == rows else ds * (float(rows) / float(ctx.count))
        L.call('rvt_bn_act_bwd_apply', L.ptr(dy), L.ptr(y0), L.ptr(fin[2]), L.ptr(fin[3]), L.ptr(fin[0]), L.ptr(fin[1]), L.ptr(dsk[0]),
               L.ptr(dsk[1]), L.ptr(dconv), L.dtype_code(dt), rows, Cout, BN_ACT_SILU, st)
        dx = None
        if ctx.needs_input_grad[0]:
            wd = ctx.pk.wd if ctx.pk is not None else weights.pack_conv_dgrad(w.detach(), s, pad, dt)
            dx = ops.conv_dgrad(dconv, wd, None, H, W, Cin, k, s, pad)
        dw = None
        if ctx.needs_input_grad[1]:
            if ctx.pk is not None:
                dwp = ctx.pk.dwp if ctx.pk.dwp_clean else ctx.pk.dwp.zero_()          # (zeroed with the whole arena at the top of the forward)
                ctx.pk.dwp_clean = False
            else:
                dwp = torch.zeros(Cout, k * k * Cin, dtype=torch.float32, device=dev)
            ops.conv_wgrad(x, dconv, dwp, k, s, pad)
            dw = weights.unpack_conv_wgrad(dwp, Cin, k).to(w.dtype)
            if ctx.pk is not None and dw.untyped_storage().data_ptr() == dwp.untyped_storage().data_ptr():
                # 1 x 1 convolutions: the unpack is a view of the ConvPack arena.  AccumulateGrad would adopt it as .grad, and the
                # arena is zeroed / rewritten by the next forward / backward (gradient accumulation, zero_grad(set_to_none=False),
                # retain_graph): hand autograd a tensor of its own
                dw = dw.clone()
        return dx, dw, ds_local[1].to(w.dtype), ds_local[0].to(w.dtype), None, None, None


class BaseConv(nn.Module):
    """Conv2d -> BatchNorm2d -> SiLU (network_blocks.py:29-53).  `conv` and `bn` are the reference's own parameter containers
    (same names, shapes, initialisation); their torch forward is never called."""

    def __init__(self, in_channels: int, out_channels: int, ksize: int, stride: int, groups: int = 1, bias: bool = False, act: str = 'silu'):
        super().__init__()
        if groups != 1 or bias or act != 'silu':
            raise NotImplementedError('rvt_amd.fpn.BaseConv: groups = 1, bias = False, act = "silu" (every shipped config)')
        self.in_channels, self.out_channels, self.ksize, self.stride, self.pad = in_channels, out_channels, ksize, stride, (ksize - 1) // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride, padding=self.pad, groups=1, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self._pk = None                                 # kernel-side weight views, set by the owning module's ConvPack (None: packed per call)

    def forward(self, x: Tensor) -> Tensor:
        """x: channels-last (N, H, W, Cin) in the compute dtype."""
        pk = self._pk
        if not self.training and not torch.is_grad_enabled() and pk is not None and pk.fin is not None and pk.dtype == x.dtype \
                and pk.cin == x.shape[-1]:
            # inference fast path: one launch, no autograd node
            return ops.conv_bn_act_fwd(x.contiguous(), pk.wp, pk.fin[2], pk.fin[3], self.ksize, self.stride, self.pad, BN_ACT_SILU)
        return _BaseConvFn.apply(x.contiguous(), self.conv.weight, self.bn.weight, self.bn.bias, self, self.training, self.training)


class _Packed:
    __slots__ = ('wp', 'wd', 'stats', 'dwp', 'dtype', 'cin', 'fin', 'dwp_clean')


class ConvPack:
    """Kernel-side weight layouts of EVERY BaseConv below a module in two flat buffers, refreshed by ONE rvt_pack_table launch per
    forward (training) or per parameter change (eval) — the same descriptor-table mechanism as the backbone's ModelWeights
    (rvt_amd/weights.py): tap-major forward weights, stride-parity input-gradient panels, plus the BatchNorm statistics and raw
    weight-gradient scratch, instead of ~10 small torch launches per convolution and step."""

    def __init__(self, root: nn.Module):
        self.convs = [m for m in root.modules() if isinstance(m, BaseConv)]
        self.sig = None
        self.versions = None

    def refresh(self, dtype: torch.dtype, training: bool) -> None:
        ps = [c.conv.weight for c in self.convs]
        if not ps or any(p.dtype != torch.float32 or not p.is_contiguous() or c.in_channels % 8 for p, c in zip(ps, self.convs)):
            for c in self.convs:
                c._pk = None
            return
        sig = (dtype, tuple((p.data_ptr(), p.device) for p in ps))
        if sig != self.sig:
            self._build(dtype, ps[0].device)
            self.sig, self.versions = sig, None
        ver = tuple(p._version for p in ps)
        if training or ver != self.versions:             # (optimizers may write through .data without bumping _version: training re-packs)
            L.call('rvt_pack_table', L.ptr(self.table.dev), len(self.table), self.table.blocks, L.dtype_code(dtype), L.stream_of(self.bufT))
            self.versions = ver
        if training:
            self.buf32.zero_()                           # batch statistics AND raw weight-gradient scratch of every unit: one memset
            if self.counters:
                torch._foreach_add_(self.counters, 1)    # num_batches_tracked of every BatchNorm
            self.bn_versions = None
            for c in self.convs:
                c._pk.fin = None
                c._pk.dwp_clean = True
        else:                                            # inference: the BatchNorm affine of every unit, once per parameter / buffer change
            bnv = tuple((t.data_ptr(), t._version) for c in self.convs for t in (c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var))
            if bnv != getattr(self, 'bn_versions', None) or any(c._pk.fin is None for c in self.convs):
                for c in self.convs:
                    bn, Cout = c.bn, c.out_channels
                    fin = torch.empty(4, Cout, dtype=torch.float32, device=bn.weight.device)
                    L.call('rvt_bn_finalize', None, None, 1, L.ptr(bn.weight.detach().float().contiguous()),
                           L.ptr(bn.bias.detach().float().contiguous()), float(bn.eps), 0.0, L.ptr(bn.running_mean), L.ptr(bn.running_var),
                           L.ptr(fin[0]), L.ptr(fin[1]), L.ptr(fin[2]), L.ptr(fin[3]), Cout, 0, L.stream_of(fin))
                    c._pk.fin = fin
                self.bn_versions = bnv

    def _build(self, dtype: torch.dtype, dev) -> None:
        arT, ar32 = weights._Arena(), weights._Arena()
        for second in (False, True):
            if second:
                arT.buf = torch.empty(max(arT.n, 64), dtype=dtype, device=dev)
                ar32.buf = torch.zeros(max(ar32.n, 64), dtype=torch.float32, device=dev)
                n_stats = ar32_stats
                arT.n = ar32.n = 0
            tab = weights._Table(weights.PACK_DT, 1024)
            for c in self.convs:                         # statistics first: one contiguous region to zero per forward
                st = ar32.take(2, c.out_channels)
                if second:
                    c._pk = _Packed()
                    c._pk.stats, c._pk.dtype, c._pk.cin, c._pk.fin, c._pk.dwp_clean = st, dtype, c.in_channels, None, False
            ar32_stats = ar32.n
            for c in self.convs:
                Cout, Cin, k, s, pad = c.out_channels, c.in_channels, c.ksize, c.stride, c.pad
                w = c.conv.weight.detach()
                wp = arT.take(Cout, k * k * Cin)
                parts, total = [], 0
                for py in range(s):
                    for px in range(s):
                        ky, kx = weights.conv_dgrad_taps(k, s, pad, py), weights.conv_dgrad_taps(k, s, pad, px)
                        parts.append((ky, kx, Cin * len(ky) * len(kx) * Cout))
                        total += parts[-1][2]
                wd = arT.take(total)
                dwp = ar32.take(Cout, k * k * Cin)
                if second:
                    weights._pack_entry(tab, w, wp, weights.PACK_CONV_FWD, (Cout, Cin, k, Cin))
                    off = 0
                    for ky, kx, cnt in parts:
                        if cnt:
                            weights._pack_entry(tab, w, wd[off:off + cnt], weights.PACK_CONV_DGRAD, (Cout, Cin, k, len(ky), len(kx)), None, ky, kx, n=cnt)
                        off += cnt
                    c._pk.wp, c._pk.wd, c._pk.dwp = wp, wd, dwp
        self.bufT, self.buf32 = arT.buf, ar32.buf
        self.stats_all = ar32.buf[:n_stats]
        self.counters = [c.bn.num_batches_tracked for c in self.convs if c.bn.track_running_stats]
        tab.upload(dev)
        self.table = tab


class Bottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act='silu'):
        super().__init__()
        if depthwise:
            raise NotImplementedError('depthwise PAFPN blocks are not built (no shipped config enables them)')
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + x if self.use_add else y


class CSPLayer(nn.Module):
    """network_blocks.py:104-141."""

    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act='silu'):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])

    def forward(self, x):
        x_1 = self.m(self.conv1(x))
        x_2 = self.conv2(x)
        return self.conv3(torch.cat((x_1, x_2), dim=-1))                  # channels-last: the channel axis is the last one


def _upsample2(x: Tensor) -> Tensor:
    """nearest-exact, scale 2 (yolo_pafpn.py:49) on a channels-last map: output pixel (y, x) <- input (y // 2, x // 2).
    One broadcast copy forward, one 2 x 2 sum backward."""
    N, H, W, C = x.shape
    return x[:, :, None, :, None, :].expand(N, H, 2, W, 2, C).reshape(N, 2 * H, 2 * W, C)


class YOLOPAFPN(nn.Module):
    def __init__(self, depth: float = 1.0, in_stages: Tuple[int, ...] = (2, 3, 4), in_channels: Tuple[int, ...] = (256, 512, 1024),
                 depthwise: bool = False, act: str = 'silu', compile_cfg: Optional[Dict] = None,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        assert len(in_stages) == len(in_channels) == 3, 'Current implementation only for 3 feature maps'
        if depthwise:
            raise NotImplementedError('depthwise PAFPN is not built (no shipped config enables it)')
        self.in_features, self.in_channels, self.compute_dtype = tuple(in_stages), tuple(in_channels), compute_dtype
        c0, c1, c2 = in_channels
        n = round(3 * depth)
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = BaseConv(c0, c0, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = BaseConv(c1, c1, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, depthwise=depthwise, act=act)
        self._pack = ConvPack(self)

    def forward(self, input: Dict[int, Tensor]):
        """input[stage]: (N, C, H, W)-shaped backbone features (channels-last views are free).  Returns the three FPN maps,
        (N, C, H, W)-shaped channels-last views like the backbone's."""
        dt = self.compute_dtype
        self._pack.refresh(dt, self.training)
        x2, x1, x0 = (input[f].permute(0, 2, 3, 1).to(dt).contiguous() for f in self.in_features)
        fpn_out0 = self.lateral_conv0(x0)
        f_out0 = self.C3_p4(torch.cat([_upsample2(fpn_out0), x1], -1))
        fpn_out1 = self.reduce_conv1(f_out0)
        pan_out2 = self.C3_p3(torch.cat([_upsample2(fpn_out1), x2], -1))
        pan_out1 = self.C3_n3(torch.cat([self.bu_conv2(pan_out2), fpn_out1], -1))
        pan_out0 = self.C3_n4(torch.cat([self.bu_conv1(pan_out1), fpn_out0], -1))
        return tuple(t.permute(0, 3, 1, 2) for t in (pan_out2, pan_out1, pan_out0))


def build_yolox_fpn(fpn_cfg, in_channels: Tuple[int, ...], compute_dtype: torch.dtype = torch.float32) -> YOLOPAFPN:
    """Registry entry point (reference yolox_extension/models/build.py:21-29)."""
    d = dict(fpn_cfg)
    name = d.pop('name')
    if name not in ('PAFPN', 'pafpn'):
        raise NotImplementedError(name)
    d.pop('compile', None)
    return YOLOPAFPN(in_channels=tuple(in_channels), compute_dtype=compute_dtype, **{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()})
