"""MI355X-native drop-in for the reference recurrent backbone.

Mirrors /root/reference/models/detection/recurrent_backbone/maxvit_rnn.py:
  * ``RNNDetector(mdl_config)`` reads the same ``model.backbone`` keys (maxvit_rnn.py:28-33,144-160;
    maxvit.py:134,157-158,201-213), exposes ``stage_dims``, ``strides``, ``num_stages``, ``stages``,
    ``get_stage_dims`` / ``get_strides`` (maxvit_rnn.py:81-91);
  * owns ``nn.Parameter``s with the reference's names and shapes, default-initialised the same way, so
    reference checkpoints load with ``load_state_dict``;
  * ``forward(x, prev_states=None, token_mask=None) -> ({1..4: (B,C,H,W)}, [(h,c)]*4)`` like
    maxvit_rnn.py:93-105, differentiable w.r.t. parameters and ``prev_states``;
  * ``forward_sequence(xs, prev_states)`` runs a whole (T,B,C,h,w) sequence stage-major (rvt_amd/stage.py) —
    what the training step (rvt_amd/step.py) calls instead of T separate ``forward`` calls.
All arithmetic happens in the HIP kernels behind include/rvt_hip.h; the torch modules below are
parameter containers only and are never called.
"""
from __future__ import annotations

import weakref

from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import ops, tuning
from .stage import SideStream, StageGeom, stage_seq_backward, stage_seq_forward, use_lstm_scan
from .weights import ModelWeights, param_signature, param_versions, round8

Tensor = torch.Tensor
LstmState = Optional[Tuple[Tensor, Tensor]]
LstmStates = List[LstmState]
BackboneFeatures = Dict[int, Tensor]


class _Holder(nn.Module):
    """Parameter container (keeps the reference's state_dict hierarchy); never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('parameter container: compute happens in rvt_amd HIP kernels')


def _cfg_get(cfg, key, default=None):
    if hasattr(cfg, 'get'):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _make_attention_block(dim: int, attention_cfg, skip_first_norm: bool) -> _Holder:
    """Parameters of PartitionAttentionCl (maxvit.py:193-250)."""
    norm_eps = _cfg_get(attention_cfg, 'norm_eps', 1e-5)
    dim_head = _cfg_get(attention_cfg, 'dim_head', 32)
    attention_bias = _cfg_get(attention_cfg, 'attention_bias', True)
    mlp_bias = _cfg_get(attention_cfg, 'mlp_bias', True)
    mlp_ratio = _cfg_get(attention_cfg, 'mlp_ratio', 4)
    ls_init = _cfg_get(attention_cfg, 'ls_init_value', 1e-5)
    if attention_cfg.use_torch_mha:
        raise NotImplementedError('use_torch_mha=True has a different parameter layout (maxvit.py:307-325)')
    if attention_cfg.mlp_gated:
        raise NotImplementedError('mlp_gated=True (GLU, maxvit.py:56-82) is not built; all shipped configs use False')
    if attention_cfg.mlp_activation != 'gelu':
        raise NotImplementedError(f'mlp_activation={attention_cfg.mlp_activation}: only exact GELU is built')
    for k in ('drop_path', 'drop_mlp'):
        if _cfg_get(attention_cfg, k, 0.0) > 0:
            raise NotImplementedError(f'{k} > 0: stochastic layers are not built (all shipped configs use 0)')
    if not (attention_bias and mlp_bias) or not ls_init > 0:
        raise NotImplementedError('bias-free linears / ls_init_value<=0 are not built')
    assert dim % dim_head == 0 and dim_head % 8 == 0 and dim_head <= 32, f'dim_head={dim_head} unsupported'
    blk = _Holder()
    blk.norm1 = nn.Identity() if skip_first_norm else nn.LayerNorm(dim, eps=norm_eps)
    blk.self_attn = _Holder()
    blk.self_attn.qkv = nn.Linear(dim, dim * 3, bias=True)
    blk.self_attn.proj = nn.Linear(dim, dim, bias=True)
    blk.ls1 = _Holder()
    blk.ls1.gamma = nn.Parameter(ls_init * torch.ones(dim))
    blk.norm2 = nn.LayerNorm(dim, eps=norm_eps)
    blk.mlp = _Holder()
    inner = int(dim * mlp_ratio)
    blk.mlp.net = nn.Sequential(nn.Sequential(nn.Linear(dim, inner, bias=True), nn.GELU()), nn.Dropout(0.0),
                                nn.Linear(inner, dim, bias=True))
    blk.ls2 = _Holder()
    blk.ls2.gamma = nn.Parameter(ls_init * torch.ones(dim))
    return blk


class RNNDetectorStage(_Holder):
    """Parameters of one stage (maxvit_rnn.py:130-167)."""

    def __init__(self, dim_in: int, stage_dim: int, spatial_downsample_factor: int, num_blocks: int,
                 enable_token_masking: bool, stage_cfg):
        super().__init__()
        assert isinstance(num_blocks, int) and num_blocks > 0
        ds, lstm_cfg, attention_cfg = stage_cfg.downsample, stage_cfg.lstm, stage_cfg.attention
        if ds.type != 'patch':
            raise NotImplementedError(ds.type)                                   # maxvit.py:134-140
        assert spatial_downsample_factor in (2, 4, 8)
        overlap = _cfg_get(ds, 'overlap', True)
        if not _cfg_get(ds, 'norm_affine', True):
            raise NotImplementedError('norm_affine=False is not built')
        k = (spatial_downsample_factor - 1) * 2 + 1 if overlap else spatial_downsample_factor
        pad = k // 2 if overlap else 0
        self.geom_conv = (k, spatial_downsample_factor, pad)
        self.downsample_cf2cl = _Holder()
        self.downsample_cf2cl.conv = nn.Conv2d(dim_in, stage_dim, k, stride=spatial_downsample_factor, padding=pad,
                                               bias=False)
        self.downsample_cf2cl.norm = nn.LayerNorm(stage_dim, eps=1e-5)
        blocks = []
        for i in range(num_blocks):
            pair = _Holder()
            pair.att_window = _make_attention_block(stage_dim, attention_cfg, skip_first_norm=(i == 0))
            pair.att_grid = _make_attention_block(stage_dim, attention_cfg, skip_first_norm=False)
            blocks.append(pair)
        self.att_blocks = nn.ModuleList(blocks)
        if _cfg_get(lstm_cfg, 'drop_cell_update', 0) > 0:
            raise NotImplementedError('drop_cell_update > 0 is not built')
        self.lstm = _Holder()
        if lstm_cfg.dws_conv:                                                    # rnn.py:24-29
            assert isinstance(lstm_cfg.dws_conv_only_hidden, bool)
            kk = lstm_cfg.dws_conv_kernel_size
            if kk != 3:
                raise NotImplementedError(f'dws_conv_kernel_size={kk}: only 3 is built')
            cg = stage_dim if lstm_cfg.dws_conv_only_hidden else stage_dim * 2
            self.lstm.conv3x3_dws = nn.Conv2d(cg, cg, kk, padding=kk // 2, groups=cg)
        self.lstm.conv1x1 = nn.Conv2d(stage_dim * 2, stage_dim * 4, kernel_size=1)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, 1, stage_dim)) if enable_token_masking else None
        if self.mask_token is not None:
            nn.init.normal_(self.mask_token, std=.02)                            # maxvit_rnn.py:166


class RNNDetector(nn.Module):
    def __init__(self, mdl_config, compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        in_channels = mdl_config.input_channels
        embed_dim = mdl_config.embed_dim
        dim_multiplier = tuple(mdl_config.dim_multiplier)
        num_blocks = tuple(mdl_config.num_blocks)
        t_max = tuple(mdl_config.T_max_chrono_init)          # parsed, asserted, unused — like the reference
        enable_masking = mdl_config.enable_masking
        num_stages = len(num_blocks)
        assert num_stages == 4
        assert isinstance(embed_dim, int)
        assert num_stages == len(dim_multiplier) == len(t_max)
        patch_size = mdl_config.stem.patch_size
        attention_cfg = mdl_config.stage.attention
        ps = attention_cfg.partition_size
        self.partition_size = (ps, ps) if isinstance(ps, int) else tuple(ps)
        assert len(self.partition_size) == 2
        self.dim_head = _cfg_get(attention_cfg, 'dim_head', 32)
        self.norm_eps = _cfg_get(attention_cfg, 'norm_eps', 1e-5)
        self.in_channels = in_channels
        self.num_blocks = num_blocks
        in_res = _cfg_get(mdl_config, 'in_res_hw', None)
        self.in_res_hw = tuple(in_res) if in_res is not None else None
        self.compute_dtype = compute_dtype

        self.stage_dims = [embed_dim * x for x in dim_multiplier]
        self.stages = nn.ModuleList()
        self.strides = []
        input_dim, stride = in_channels, 1
        for si in range(num_stages):
            f = patch_size if si == 0 else 2
            self.stages.append(RNNDetectorStage(input_dim, self.stage_dims[si], f, num_blocks[si],
                                                enable_masking and si == 0, mdl_config.stage))
            stride *= f
            self.strides.append(stride)
            input_dim = self.stage_dims[si]
        self.num_stages = num_stages
        self._param_names = [n for n, _ in self.named_parameters()]

    # ---- reference API ------------------------------------------------------------------------------
    def get_stage_dims(self, stages: Tuple[int, ...]) -> Tuple[int, ...]:
        idx = [x - 1 for x in stages]
        assert min(idx) >= 0 and max(idx) < len(self.stages), idx
        return tuple(self.stage_dims[i] for i in idx)

    def get_strides(self, stages: Tuple[int, ...]) -> Tuple[int, ...]:
        idx = [x - 1 for x in stages]
        assert min(idx) >= 0 and max(idx) < len(self.stages), idx
        return tuple(self.strides[i] for i in idx)

    def forward(self, x: Tensor, prev_states: Optional[LstmStates] = None, token_mask: Optional[Tensor] = None) \
            -> Tuple[BackboneFeatures, LstmStates]:
        feats, states = self.forward_sequence(x[None], prev_states, None if token_mask is None else token_mask[None])
        return {k: v[0] for k, v in feats.items()}, states

    # ---- sequence API -------------------------------------------------------------------------------
    def stage_geoms(self, H: int, W: int) -> List[StageGeom]:
        out = []
        cin = self.in_channels
        for si, st in enumerate(self.stages):
            k, s, p = st.geom_conv
            g = StageGeom(C=self.stage_dims[si], Cin=cin, H_in=H, W_in=W, k=k, stride=s, pad=p,
                          ph=self.partition_size[0], pw=self.partition_size[1], dim_head=self.dim_head,
                          num_blocks=self.num_blocks[si], eps=self.norm_eps)
            H, W, cin = g.H, g.W, g.C
            assert H % g.ph == 0 and W % g.pw == 0, \
                f'stage {si + 1}: {H}x{W} not divisible by partition {self.partition_size} (maxvit.py:275-276)'
            out.append(g)
        return out

    def forward_sequence(self, xs: Union[Tensor, Sequence[Tensor]], prev_states: Optional[LstmStates] = None,
                         token_masks: Optional[Tensor] = None):
        """xs: (T,B,Cin,h,w) uint8/float or a list of T (B,Cin,h,w) tensors (the reference's EV_REPR list).
        Returns ({stage: (T,B,C,H,W)}, [(h,c)]*4); h = features of the last step, c fp32."""
        if not torch.is_tensor(xs):
            xs = torch.stack(list(xs), 0)
        assert xs.dim() == 5 and xs.shape[2] == self.in_channels
        if prev_states is None:
            prev_states = [None] * self.num_stages
        assert len(prev_states) == self.num_stages
        flat_states = []
        for st in prev_states:
            flat_states += [None, None] if st is None else [st[0], st[1]]
        params = [p for _, p in self.named_parameters()]
        if not torch.is_grad_enabled() and tuning.get('route_stage_driver') != 0:
            # validation / streaming inference: no autograd node, one library call per stage (csrc/capi_stage.hip)
            r = self._forward_nograd(xs, prev_states, token_masks, params)
            if r is not None:
                return r
        # grad mode is read HERE: inside autograd.Function.forward it is always off, and needs_input_grad stays True for
        # parameters under torch.no_grad() — validation / streaming inference must not take the training path
        outs = _BackboneSeqFn.apply(self, xs, token_masks, torch.is_grad_enabled(), *flat_states, *params)
        feats = {s + 1: outs[2 * s] for s in range(self.num_stages)}
        # the returned state tensors are the caller's to keep (RNNStates stores them per worker) and to mutate in place
        # (masked reset): `c` is a buffer of its own; `h` is cloned off the (T+1)-slot feature buffer when the caller
        # is not going to differentiate through it, so that a kept state does not pin the whole sequence
        states = []
        for s in range(self.num_stages):
            h, c = outs[2 * s][-1], outs[2 * s + 1]
            states.append((h if h.requires_grad else h.clone(memory_format=torch.preserve_format), c))
        return feats, states

    def _forward_nograd(self, xs: Tensor, prev_states, token_masks, params):
        """The no-grad forward through the C-side stage driver (include/rvt_hip.h: rvt_stage_seq_fwd): Python allocates, the library
        sequences the kernels.  Returns None when a stage is outside the driver's coverage (token masks, DWS-ConvLSTM): the caller
        then takes the operator-by-operator host loop of rvt_amd/stage.py (same kernels)."""
        from . import stage_driver as SD
        ns, dt = self.num_stages, self.compute_dtype
        T, B, Cin, h, w = xs.shape
        Hm, Wm = self.in_res_hw if self.in_res_hw is not None else (h, w)
        assert h <= Hm and w <= Wm, f'input {h}x{w} larger than model resolution {Hm}x{Wm}'
        geoms = self.stage_geoms(Hm, Wm)
        mw = self.model_weights(params, geoms, dt, False)
        if token_masks is not None or any(sw.dws is not None for sw in mw.stages):
            return None
        src = xs.reshape(T * B, Cin, h, w)
        if src.dtype not in (torch.uint8, torch.float32):
            src = src.float()
        u8 = src.dtype == torch.uint8
        if u8:
            inp = src.contiguous()
        else:
            inp = ops.prepack_input(src, Hm, Wm, round8(Cin), dt)
        key = (dt, u8, h, w, Hm, Wm)
        calls = getattr(mw, '_stage_calls', None)
        if calls is None or calls[0] != key:
            calls = (key, [SD.StageCall(mw.stages[si], geoms[si], dt, u8 and si == 0, h, w) for si in range(ns)])
            mw._stage_calls = calls
        feats, states = {}, []
        for si in range(ns):
            st = prev_states[si]
            h0 = c0 = None
            if st is not None:
                h0, c0 = _to_cl(st[0], dt), _to_cl(st[1], torch.float32)
            Hall, c_last = SD.stage_seq_fwd(calls[1][si], inp, h0, c0, T, B)
            g = geoms[si]
            inp = Hall[1:].reshape(T * B, g.H, g.W, g.C)
            feats[si + 1] = Hall[1:].permute(0, 1, 4, 2, 3)
            states.append((feats[si + 1][-1].clone(memory_format=torch.preserve_format), c_last.permute(0, 3, 1, 2)))
        self._last_saved = None      # (test hook, see _BackboneSeqFn.forward: nothing is kept on this route)
        return feats, states

    def invalidate_weight_cache(self) -> None:
        """Force a rebuild of the kernel-side weight copies on the next forward (call after writing parameters through
        ``.data`` in inference code; training forwards re-pack every step anyway)."""
        self._mw_cache = None

    def model_weights(self, params, geoms, dt: torch.dtype, need_grad: bool) -> ModelWeights:
        sig = (dt, param_signature(params))
        mw = getattr(self, '_mw_cache', None)
        fresh = mw is None or mw.sig != sig or (need_grad and not mw.need_grad)
        if fresh:
            mw = ModelWeights(self, dict(zip(self._param_names, params)), geoms, dt, need_grad)
            mw.sig, mw.versions = sig, None
            self._mw_cache = mw
        ver = param_versions(params)
        # training: re-pack every step (optimizers may write through .data, which does not bump _version: one launch);
        # inference: only when a parameter changed (streaming inference calls forward once per time step)
        if fresh or need_grad or mw.versions != ver:
            mw.pack()
            mw.versions = ver
        return mw


def _to_cl(t: Tensor, dtype: torch.dtype) -> Tensor:
    """(…,C,H,W)-shaped -> contiguous (…,H,W,C) in `dtype` (free when t is already a channels-last view)."""
    nd = t.dim()
    perm = list(range(nd - 3)) + [nd - 2, nd - 1, nd - 3]
    return t.permute(*perm).to(dtype).contiguous()


class _BackboneSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: RNNDetector, xs: Tensor, token_masks: Optional[Tensor], grad_enabled: bool, *rest):
        ns = mod.num_stages
        states_in, params = rest[:2 * ns], rest[2 * ns:]
        names = mod._param_names
        p = dict(zip(names, params))
        dt = mod.compute_dtype
        T, B, Cin, h, w = xs.shape
        Hm, Wm = mod.in_res_hw if mod.in_res_hw is not None else (h, w)
        assert h <= Hm and w <= Wm, f'input {h}x{w} larger than model resolution {Hm}x{Wm}'
        geoms = mod.stage_geoms(Hm, Wm)
        need_grad = bool(grad_enabled) and any(ctx.needs_input_grad[4:])
        src = xs.reshape(T * B, Cin, h, w)
        if src.dtype not in (torch.uint8, torch.float32):
            src = src.float()
        g0 = geoms[0]
        if ops.stem_supported(src, dt, g0.C, g0.k, g0.stride, g0.pad):
            inp = src.contiguous()           # the stem kernels read the planes themselves (no channels-last copy)
        else:
            inp = ops.prepack_input(src, Hm, Wm, round8(Cin), dt)
        mw = mod.model_weights(params, geoms, dt, need_grad)
        svs, outs = [], []
        for si in range(ns):
            g = geoms[si]
            pre = f'stages.{si}.'
            sw = mw.stages[si]
            h0, c0 = states_in[2 * si], states_in[2 * si + 1]
            if h0 is not None:
                h0, c0 = _to_cl(h0, dt), _to_cl(c0, torch.float32)
            tm = mt = None
            if si == 0 and token_masks is not None:
                assert mod.stages[0].mask_token is not None, 'No mask token present in this stage'
                tm, mt = token_masks.to(xs.device), p[pre + 'mask_token'].detach()
            Hall, c_last, sv = stage_seq_forward(sw, g, inp, h0, c0, T, B, need_grad, tm, mt)
            svs.append(sv)
            inp = Hall[1:].reshape(T * B, g.H, g.W, g.C)
            outs += [Hall[1:].permute(0, 1, 4, 2, 3), c_last.permute(0, 3, 1, 2)]
        ctx.mod, ctx.geoms, ctx.mw, ctx.svs, ctx.p = mod, geoms, mw, svs, p
        ctx.T, ctx.B = T, B
        ctx.set_materialize_grads(False)
        # (test hook: weak reference only - a forward whose backward never runs must not pin its activations on the module)
        mod._last_saved = weakref.ref(svs[0]) if need_grad and svs and svs[0] is not None else None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gout):
        mod, geoms, T, B, mw = ctx.mod, ctx.geoms, ctx.T, ctx.B, ctx.mw
        ns = mod.num_stages
        dt = mod.compute_dtype
        p = ctx.p
        state_grads: List[Optional[Tensor]] = [None] * (2 * ns)
        d_from_above = None          # conv dgrad of stage s+1, already including this stage's own feature cotangent
        hook = getattr(mod, '_stage_grad_hook', None)
        # Gradient buckets: a parameter whose .grad already IS its bucket view (the caller did not zero it) keeps
        # accumulating — every kernel adds into the bucket — otherwise the stage's bucket starts from zero.
        accumulate = []
        for si in range(ns):
            sg = mw.grads[si]
            mine = [p[n].grad is not None and p[n].grad.data_ptr() == sg.g(n).data_ptr() for n in sg.names]
            acc = all(mine)
            accumulate.append(acc)
            if not acc:
                for n, m_ in zip(sg.names, mine):     # mixed: detach the accumulated ones from the bucket first
                    if m_:
                        p[n].grad = p[n].grad.clone()
            sg.zero(keep_params=acc)
        # weight-gradient GEMMs run on a side stream, joined at the end of every stage (measured: deferring the join to
        # the end of backward is slower — 162 vs 145 ms/step — the caching allocator cannot recycle the operands the side
        # stream still holds and the extra resident work competes with the next stage's critical path)
        side = SideStream(ctx.svs[ns - 1].y0)
        held_hooks: List[int] = []      # deferred mode: stages whose weight gradients are still queued / running on the side stream

        def call_hook(sj: int) -> None:
            if hook is None:
                return
            try:                        # e.g. rvt_amd.dist.StageGradReducer: start this stage's all-reduce now
                hook(sj, mw.grads[sj].param_region, accumulated=accumulate[sj])
            except TypeError as e:      # a user-installed two-argument hook (the pre-round-3 signature)
                if 'accumulated' not in str(e):
                    raise
                hook(sj, mw.grads[sj].param_region)
        for si in range(ns - 1, -1, -1):
            g = geoms[si]
            # deferred weight gradients (tuning.route_wgrad_stream = 2): those of the stage above start now, beside this stage's reverse
            # scan; this stage queues its own iff the stage below scans per step (a chip-filling scan kernel leaves nothing to fill)
            side.flush()
            side.deferring = side.defer_mode and si > 0 and not use_lstm_scan(dt, geoms[si - 1].C, mw.stages[si - 1].dws, T, True)
            dF, dC = gout[2 * si], gout[2 * si + 1]
            if si == ns - 1:
                dH = None if dF is None else _to_cl(dF, dt)
            else:
                dH = d_from_above.view(T, B, g.H, g.W, g.C)
            dc_last = None if dC is None else _to_cl(dC, torch.float32)
            # cotangent attached to THIS stage's input frames = feature cotangent of the stage below
            prev_cot = None
            if si > 0 and gout[2 * (si - 1)] is not None:
                gp = geoms[si - 1]
                prev_cot = _to_cl(gout[2 * (si - 1)], dt).view(T * B, gp.H, gp.W, gp.C)
            d_in, dh0, dc0 = stage_seq_backward(mw.stages[si], g, ctx.svs[si], dH, dc_last, T, B, si > 0, prev_cot,
                                                mw.grads[si], f'stages.{si}.', side=side,
                                                finalize=lambda si=si: mw.finalize_stage_grads(si))
            d_from_above = d_in
            for sj in held_hooks:         # (stage_seq_backward ended with side.join(): the stage above is complete now)
                call_hook(sj)
            held_hooks = []
            if side.deferring:
                held_hooks.append(si)
            else:
                call_hook(si)
            if ctx.needs_input_grad[4 + 2 * si]:
                state_grads[2 * si] = dh0.permute(0, 3, 1, 2)
                state_grads[2 * si + 1] = dc0.permute(0, 3, 1, 2)
            ctx.svs[si] = None
        if side.flush():                  # (cannot happen with the rule above - stage 1 never defers - but a queue must never be dropped)
            side.join()
        for sj in held_hooks:
            call_hook(sj)
        finish = getattr(mod, '_stage_grad_finish', None)
        if finish is not None:            # data parallel: order everything downstream after the per-stage all-reduces
            finish()
        # Hand-off.  Default: the gradients go back through autograd as COPIES of the bucket views, so every autograd contract
        # holds (torch.autograd.grad returns them, a p.grad kept across steps is never overwritten, AccumulateGrad may steal the
        # tensor).  One pass over 51 MB of fp32 at RVT-Base: ~0.03 ms.  With mod.zero_copy_grads = True (bench.py, GraphedStep:
        # stable addresses for a captured hipGraph and the fused optimizer) the parameter's .grad BECOMES the persistent bucket
        # view instead and None is returned to autograd: then the next backward rewrites that memory, and
        # torch.autograd.grad(...) sees no gradient for the backbone parameters.
        zero_copy = getattr(mod, 'zero_copy_grads', False)
        pgrads = []
        for i, n in enumerate(mod._param_names):
            if not ctx.needs_input_grad[4 + 2 * ns + i]:
                pgrads.append(None)
                continue
            si = int(n.split('.')[1])
            view = mw.grads[si].g(n)
            if view.dtype != p[n].dtype:
                view = view.to(p[n].dtype)
            if not zero_copy:
                pgrads.append(view.clone())
                continue
            pgrads.append(None)
            if p[n].grad is None:
                p[n].grad = view
            elif p[n].grad.data_ptr() != view.data_ptr():
                p[n].grad.add_(view)
        return (None, None, None, None, *state_grads, *pgrads)


def build_recurrent_backbone(backbone_cfg, compute_dtype: torch.dtype = torch.float32):
    """Registry entry point (reference recurrent_backbone/__init__.py:6-11)."""
    if backbone_cfg.name == 'MaxViTRNN':
        return RNNDetector(backbone_cfg, compute_dtype=compute_dtype)
    raise NotImplementedError(backbone_cfg.name)
