"""Kernel-side views of the module parameters and the gradient buckets behind them.

The nn.Module keeps the reference's parameter names and shapes as fp32 masters; the HIP kernels want
  * conv weights tap-major / channels-last, input channels zero-padded to a multiple of 8,
  * transposed copies for the input-gradient GEMMs (LayerScale gamma folded in where it applies),
  * the ConvLSTM 1x1-conv rows interleaved so one 32-column epilogue unit holds all four gates,
  * everything cast to the activation dtype.
All of that is ONE launch per optimisation step: ``ModelWeights`` lays the packed tensors out in two flat buffers,
builds a descriptor table (rvt_amd/csrc/pack.hpp) once, and ``pack()`` replays it through ``rvt_pack_table``.

Gradients go the other way through persistent per-stage fp32 buckets (``StageGrads``): the weight-gradient kernels
accumulate straight into views of a bucket (param order = the module's), the LayerScale fold and the conv unpack are one
table launch each, and the same flat region is what the data-parallel all-reduce sends (rvt_amd/dist.py) — no per-step
allocation, no concatenation.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

Tensor = torch.Tensor

PACK_COPY, PACK_TRANSPOSE, PACK_CONV_FWD, PACK_CONV_DGRAD, PACK_LSTM_ROWS, PACK_CONV_WGRAD_ACC, PACK_CONV_DGRAD4 = range(7)
PACK_DT = np.dtype([('src', '<u8'), ('dst', '<u8'), ('scale', '<u8'), ('n', '<i8'), ('kind', '<i4'), ('out_f32', '<i4'),
                    ('d', '<i4', (5,)), ('ky', '<i4', (4,)), ('kx', '<i4', (4,)), ('block0', '<u4')])
LS_DT = np.dtype([('S', '<u8'), ('cs', '<u8'), ('W', '<u8'), ('b', '<u8'), ('gamma', '<u8'), ('dW', '<u8'), ('db', '<u8'),
                  ('dgamma', '<u8'), ('C', '<i4'), ('K', '<i4'), ('block0', '<u4'), ('pad', '<i4')])
assert PACK_DT.itemsize == 96 and LS_DT.itemsize == 80


def round8(c: int) -> int:
    return (c + 7) // 8 * 8


def conv_dgrad_taps(k: int, stride: int, pad: int, parity: int) -> List[int]:
    return [t for t in range(k) if t % stride == (parity + pad) % stride]


class _Table:
    """Host-built descriptor array -> device bytes (one launch walks it)."""

    def __init__(self, dt: np.dtype, per_block: Optional[int]):
        self.dt, self.per_block = dt, per_block
        self.rows: List[dict] = []
        self.blocks = 0
        self.dev: Optional[Tensor] = None

    def add(self, nblocks: int, **fields) -> None:
        fields['block0'] = self.blocks
        self.rows.append(fields)
        self.blocks += nblocks

    def upload(self, device) -> None:
        arr = np.zeros(len(self.rows), dtype=self.dt)
        for i, r in enumerate(self.rows):
            for k, v in r.items():
                arr[i][k] = v
        self.dev = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(device)

    def __len__(self):
        return len(self.rows)


def _pack_entry(tab: _Table, src: Tensor, dst: Tensor, kind: int, d=(), scale: Optional[Tensor] = None, ky=(), kx=(),
                n: Optional[int] = None) -> None:
    n = dst.numel() if n is None else n
    dd = list(d) + [0] * (5 - len(d))
    tab.add((n + 1023) // 1024, src=src.data_ptr(), dst=dst.data_ptr(), scale=0 if scale is None else scale.data_ptr(), n=n,
            kind=kind, out_f32=int(dst.dtype == torch.float32), d=dd, ky=list(ky) + [0] * (4 - len(ky)),
            kx=list(kx) + [0] * (4 - len(kx)))


class _Arena:
    """Two-pass bump allocator over one flat tensor: first pass sizes, second pass hands out views."""

    def __init__(self):
        self.n = 0
        self.buf: Optional[Tensor] = None

    def take(self, *shape) -> Optional[Tensor]:
        cnt = int(np.prod(shape))
        off = self.n
        self.n += (cnt + 63) // 64 * 64            # keep every view 128/256-byte aligned
        if self.buf is None:
            return None
        return self.buf[off:off + cnt].view(*shape)


class StageWeights:
    """Everything one stage's kernels read (views into ModelWeights' flat buffers, or the fp32 parameters themselves)."""
    pack_epoch = 0            # bumped by ModelWeights.pack(): derived copies made lazily below are stale afterwards
    _scan3 = None

    def scan3_packed(self, bwd: bool) -> Tensor:
        """The ConvLSTM weights in the operand order the wide-stage scan kernels stream (csrc/lstm_scan3.hpp), re-derived from
        `lstm_wn` once per pack() and direction; the buffers persist (stable addresses for hipGraph replay)."""
        from . import ops
        st = self._scan3
        if st is None:
            st = self._scan3 = dict(fwd=None, bwd=None, fwd_epoch=-1, bwd_epoch=-1)
        key = 'bwd' if bwd else 'fwd'
        if st[key] is None:
            st[key] = torch.empty(self.lstm_wn.numel(), dtype=self.lstm_wn.dtype, device=self.lstm_wn.device)
        if st[key + '_epoch'] != self.pack_epoch:
            L.call('rvt_lstm_scan3_pack', L.ptr(self.lstm_wn), None if bwd else L.ptr(st['fwd']), L.ptr(st['bwd']) if bwd else None,
                   self.lstm_wn.shape[1] // 2, L.stream_of(self.lstm_wn))
            st[key + '_epoch'] = self.pack_epoch
        return st[key]


class StageGrads:
    """fp32 gradient bucket of one stage: [parameter gradients in module order | raw products that still need a fold]."""

    def __init__(self, names: List[str], shapes: Dict[str, Tuple[int, ...]], aux: List[Tuple[str, Tuple[int, ...]]], device):
        self.names = names
        off = 0
        spans = {}
        for n in names:
            cnt = int(np.prod(shapes[n]))
            spans[n] = (off, cnt, shapes[n])
            off += (cnt + 3) // 4 * 4               # 16-byte aligned views
        self.n_param = off
        for n, shp in aux:
            cnt = int(np.prod(shp))
            spans[n] = (off, cnt, shp)
            off += (cnt + 3) // 4 * 4
        self.flat = torch.zeros(max(off, 4), dtype=torch.float32, device=device)
        self.view = {n: self.flat[o:o + c].view(*shp) for n, (o, c, shp) in spans.items()}
        self.param_region = self.flat[:self.n_param]
        self.ls_table: Optional[_Table] = None
        self.unpack_table: Optional[_Table] = None

    def zero(self, keep_params: bool = False) -> None:
        """Start a backward: the raw products always restart from zero (their fold ADDS them into the parameter
        gradients); the parameter gradients themselves only when the caller is not accumulating."""
        (self.flat[self.n_param:] if keep_params else self.flat).zero_()

    def g(self, name: str) -> Tensor:
        return self.view[name]


class ModelWeights:
    def __init__(self, mod, p: Dict[str, Tensor], geoms, dtype: torch.dtype, need_grad: bool):
        self.dtype, self.need_grad = dtype, need_grad
        dev = next(iter(p.values())).device
        self.device = dev
        self._refresh: List[Tuple[Tensor, Tensor]] = []          # (fp32 contiguous shadow, parameter) for odd masters
        self._p = p
        src_cache: Dict[str, Tensor] = {}

        def master(name: str) -> Tensor:
            """fp32 contiguous view of a parameter (the parameter itself in the normal case)."""
            if name not in src_cache:
                t = p[name].detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    shadow = t.float().contiguous()
                    self._refresh.append((shadow, p[name]))
                    t = shadow
                src_cache[name] = t
            return src_cache[name]
        self.master = master

        arT, ar32 = _Arena(), _Arena()
        for second in (False, True):
            if second:
                arT.buf = torch.empty(max(arT.n, 64), dtype=dtype, device=dev)
                ar32.buf = torch.empty(max(ar32.n, 64), dtype=torch.float32, device=dev)
                arT.n = ar32.n = 0
            tab = _Table(PACK_DT, 1024)
            stages = []
            for si, g in enumerate(geoms):
                stages.append(self._build_stage(tab if second else None, arT, ar32, p, f'stages.{si}.', g, need_grad))
        self.bufT, self.buf32 = arT.buf, ar32.buf
        self.stages: List[StageWeights] = stages
        tab.upload(dev)
        self.table = tab
        self.grads: List[StageGrads] = []
        if need_grad:
            for si, g in enumerate(geoms):
                self.grads.append(self._build_grads(mod, p, f'stages.{si}.', g, stages[si], dev))

    # ---- packed weights ---------------------------------------------------------------------------------
    def _build_stage(self, tab: Optional[_Table], arT: _Arena, ar32: _Arena, p, pre: str, g, need_grad: bool) -> StageWeights:
        m = self.master
        dtype_is_bf16 = self.dtype == torch.bfloat16
        sw = StageWeights()
        C, Cin, k, stride, pad = g.C, g.Cin, g.k, g.stride, g.pad
        sw.C, sw.Cin, sw.cin_pad = C, Cin, round8(Cin)
        cp = sw.cin_pad

        def emit(src_name, dst, kind, d=(), scale=None, ky=(), kx=()):
            if tab is not None:
                _pack_entry(tab, m(src_name), dst, kind, d, None if scale is None else m(scale), ky, kx)

        wname = pre + 'downsample_cf2cl.conv.weight'
        sw.conv_w = arT.take(C, k * k * cp)
        emit(wname, sw.conv_w, PACK_CONV_FWD, (C, Cin, k, cp))
        sw.conv_wd = None
        if need_grad and Cin % 8 == 0:
            # concatenation over parity classes (py,px) of [Cin][(a,b,cout)] = w[cout,cin,Ky[a],Kx[b]] — the layout
            # rvt_conv_dgrad walks (rvt_amd/csrc/capi.hip)
            parts = []
            total = 0
            for py in range(stride):
                for px in range(stride):
                    ky, kx = conv_dgrad_taps(k, stride, pad, py), conv_dgrad_taps(k, stride, pad, px)
                    parts.append((ky, kx, Cin * len(ky) * len(kx) * C))
                    total += parts[-1][2]
            sw.conv_wd = arT.take(total)
            off = 0
            for ky, kx, cnt in parts:
                if cnt and tab is not None:
                    _pack_entry(tab, m(wname), sw.conv_wd[off:off + cnt], PACK_CONV_DGRAD, (C, Cin, k, len(ky), len(kx)),
                                None, ky, kx, n=cnt)
                off += cnt
        # ... and the block-sparse [4 Cin][4 Cout] layout of the one-launch input gradient (rvt_conv_dgrad4; 3x3 / 2 / 1 convs)
        sw.conv_wd4 = None
        if need_grad and k == 3 and stride == 2 and pad == 1 and Cin % 64 == 0 and C % 64 == 0 and dtype_is_bf16:
            sw.conv_wd4 = arT.take(4 * Cin, 4 * C)
            emit(wname, sw.conv_wd4, PACK_CONV_DGRAD4, (C, Cin))
        sw.ln_w, sw.ln_b = (m(pre + 'downsample_cf2cl.norm.weight'), m(pre + 'downsample_cf2cl.norm.bias')) if tab is not None else (None, None)
        sw.blocks = []
        for bi in range(g.num_blocks):
            pair = []
            for blk in ('att_window', 'att_grid'):
                bp = f'{pre}att_blocks.{bi}.{blk}.'
                d = {}
                has_n1 = (bp + 'norm1.weight') in p
                if tab is not None:
                    d['n1_w'] = m(bp + 'norm1.weight') if has_n1 else None
                    d['n1_b'] = m(bp + 'norm1.bias') if has_n1 else None
                    d.update(qkv_b=m(bp + 'self_attn.qkv.bias'), proj_b=m(bp + 'self_attn.proj.bias'), g1=m(bp + 'ls1.gamma'),
                             g2=m(bp + 'ls2.gamma'), n2_w=m(bp + 'norm2.weight'), n2_b=m(bp + 'norm2.bias'),
                             fc1_b=m(bp + 'mlp.net.0.0.bias'), fc2_b=m(bp + 'mlp.net.2.bias'),
                             proj_w32=m(bp + 'self_attn.proj.weight'), fc2_w32=m(bp + 'mlp.net.2.weight'))
                for key, name, shp in (('qkv_w', 'self_attn.qkv.weight', (3 * C, C)), ('proj_w', 'self_attn.proj.weight', (C, C)),
                                       ('fc1_w', 'mlp.net.0.0.weight', (4 * C, C)), ('fc2_w', 'mlp.net.2.weight', (C, 4 * C))):
                    d[key] = arT.take(*shp)
                    emit(bp + name, d[key], PACK_COPY)
                if need_grad:
                    # dgrad operands: W^T, with the LayerScale of the branch folded into proj / fc2
                    for key, name, shp, sc in (('qkv_wt', 'self_attn.qkv.weight', (3 * C, C), None),
                                               ('proj_wt', 'self_attn.proj.weight', (C, C), bp + 'ls1.gamma'),
                                               ('fc1_wt', 'mlp.net.0.0.weight', (4 * C, C), None),
                                               ('fc2_wt', 'mlp.net.2.weight', (C, 4 * C), bp + 'ls2.gamma')):
                        d[key] = arT.take(shp[1], shp[0])
                        emit(bp + name, d[key], PACK_TRANSPOSE, shp, sc)
                pair.append(d)
            sw.blocks.append(tuple(pair))
        # optional depth-wise 3x3 of the DWS-ConvLSTM (rnn.py:25-29): fp32 [Cg][k*k] + bias, Cg = C (hidden only) or 2C
        sw.dws = None
        if (pre + 'lstm.conv3x3_dws.weight') in p and tab is not None:
            wd = m(pre + 'lstm.conv3x3_dws.weight')
            cg, kk = wd.shape[0], wd.shape[-1]
            sw.dws = dict(only_hidden=(cg == C), k=kk, w=wd.view(cg, kk * kk), b=m(pre + 'lstm.conv3x3_dws.bias'))
        ln = pre + 'lstm.conv1x1.weight'
        sw.lstm_w = arT.take(4 * C, 2 * C)                       # gate-interleaved rows (per-step GEMM epilogue)
        emit(ln, sw.lstm_w, PACK_LSTM_ROWS, (C, 2 * C))
        sw.lstm_b = ar32.take(4 * C)
        emit(pre + 'lstm.conv1x1.bias', sw.lstm_b, PACK_LSTM_ROWS, (C, 1))
        sw.lstm_wn = arT.take(4 * C, 2 * C)                      # natural row order [f|i|o|g] (scan kernels)
        emit(ln, sw.lstm_wn, PACK_COPY)
        sw.lstm_bn = m(pre + 'lstm.conv1x1.bias') if tab is not None else None
        sw.lstm_wt = None
        if need_grad:
            sw.lstm_wt = arT.take(2 * C, 4 * C)
            emit(ln, sw.lstm_wt, PACK_TRANSPOSE, (4 * C, 2 * C))
        return sw

    # ---- gradient buckets ---------------------------------------------------------------------------------
    def _build_grads(self, mod, p, pre: str, g, sw: StageWeights, dev) -> StageGrads:
        names = [n for n in mod._param_names if n.startswith(pre)]
        shapes = {n: tuple(p[n].shape) for n in names}
        C, k, cp = g.C, g.k, sw.cin_pad
        aux = [('raw/conv', (C, k * k * cp))]
        for bi in range(g.num_blocks):
            for blk in ('att_window', 'att_grid'):
                bp = f'{pre}att_blocks.{bi}.{blk}.'
                aux += [(bp + 'S1', (C, C)), (bp + 'cs1', (C,)), (bp + 'S2', (C, 4 * C)), (bp + 'cs2', (C,))]
        sg = StageGrads(names, shapes, aux, dev)
        ls = _Table(LS_DT, None)
        for bi in range(g.num_blocks):
            for wi, blk in enumerate(('att_window', 'att_grid')):
                bp = f'{pre}att_blocks.{bi}.{blk}.'
                bw = sw.blocks[bi][wi]
                for S, cs, Wn, bn, gn, K in ((bp + 'S1', bp + 'cs1', 'self_attn.proj.weight', 'self_attn.proj.bias', 'ls1.gamma', C),
                                             (bp + 'S2', bp + 'cs2', 'mlp.net.2.weight', 'mlp.net.2.bias', 'ls2.gamma', 4 * C)):
                    ls.add(C, S=sg.g(S).data_ptr(), cs=sg.g(cs).data_ptr(), W=self.master(bp + Wn).data_ptr(),
                           b=self.master(bp + bn).data_ptr(), gamma=self.master(bp + gn).data_ptr(),
                           dW=sg.g(bp + Wn).data_ptr(), db=sg.g(bp + bn).data_ptr(), dgamma=sg.g(bp + gn).data_ptr(), C=C, K=K, pad=0)
        ls.upload(dev)
        sg.ls_table = ls
        up = _Table(PACK_DT, 1024)
        cw = sg.g(pre + 'downsample_cf2cl.conv.weight')
        _pack_entry(up, sg.g('raw/conv'), cw, PACK_CONV_WGRAD_ACC, (C, g.Cin, k, cp))
        up.upload(dev)
        sg.unpack_table = up
        return sg

    # ---- per-step work --------------------------------------------------------------------------------------
    def pack(self) -> None:
        """fp32 masters -> kernel-side layouts: one launch (plus a copy per non-fp32 master, normally none)."""
        for shadow, param in self._refresh:
            shadow.copy_(param.detach())
        L.call('rvt_pack_table', L.ptr(self.table.dev), len(self.table), self.table.blocks, L.dtype_code(self.dtype),
               L.stream_of(self.bufT))
        for sw in self.stages:
            sw.pack_epoch += 1

    def finalize_stage_grads(self, si: int) -> None:
        """LayerScale fold + conv weight-gradient unpack of stage si: two launches on the current stream."""
        sg = self.grads[si]
        L.call('rvt_layerscale_grad_table', L.ptr(sg.ls_table.dev), len(sg.ls_table), sg.ls_table.blocks, L.stream_of(sg.flat))
        L.call('rvt_pack_table', L.ptr(sg.unpack_table.dev), len(sg.unpack_table), sg.unpack_table.blocks, L.RVT_F32,
               L.stream_of(sg.flat))


def param_signature(params) -> tuple:
    return tuple((t.data_ptr(), t.dtype, t.device) for t in params)


def param_versions(params) -> tuple:
    return tuple(t._version for t in params)


# ---- reference implementations of the packings (tests compare the table kernel against these) ------------------
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


def pack_conv_dgrad(w: Tensor, stride: int, pad: int, dtype: torch.dtype) -> Tensor:
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


def pack_conv_dgrad4(w: Tensor, dtype: torch.dtype) -> Tensor:
    """[4*Cin][4*Cout] block-sparse weights of the one-launch input gradient (rvt_conv_dgrad4; 3x3 / stride 2 / pad 1):
    row (py, px, ci), column (da, db, co) = w[co][ci][ky][kx] with ky = 1 | 2, 0 for py = 0 | 1 and da = 0, 1 (kx likewise), else 0.
    Host-side restatement of PACK_CONV_DGRAD4 (csrc/pack.hpp) for the kernel tests."""
    Cout, Cin, k, _ = w.shape
    assert k == 3
    out = torch.zeros(4, Cin, 4, Cout, dtype=torch.float32, device=w.device)
    for py in range(2):
        for px in range(2):
            for da in range(py + 1):
                for db in range(px + 1):
                    ky = 1 if py == 0 else (2 if da == 0 else 0)
                    kx = 1 if px == 0 else (2 if db == 0 else 0)
                    out[2 * py + px, :, 2 * da + db, :] = w[:, :, ky, kx].t()
    return out.reshape(4 * Cin, 4 * Cout).to(dtype).contiguous()
