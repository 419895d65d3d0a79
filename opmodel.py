"""Algorithmic FLOP / HBM-byte model of EVERY C-ABI entry point (include/rvt_hip.h), for bench.py's `roofline` objects.

MEASUREMENT INFRASTRUCTURE (not part of the product path).  Round 3's bench could only price GEMM-family launches, so its
`roofline` could never name the worst kernels.  `model(name, args)` returns, from the launch's own
arguments (positions per include/rvt_hip.h):

  flops  ALGORITHMIC MFMA-side work of the operator, 2*MAC: the products the reference's forward / autograd
         performs for the same result - e.g. 16 M C^2 for each of {MLP forward, its input gradients, its weight gradients}.
         Products a recompute-style backward re-does (fc1 / GELU recomputed from xmid, q / k / v / P recomputed from the block
         input, ConvLSTM gates recomputed in the reverse scan) are NOT counted here: `executed(name, args)` returns the
         recompute-inclusive figure separately (round 5; the round-4 line priced the recompute as algorithmic work and
         printed 0.20 where 0.16 was right).  Element-wise work (LayerNorm, softmax, GELU, gates) is not counted.
  bytes  algorithmic HBM bytes: every operand and result tensor crosses HBM exactly once (weights once per launch);
         intermediates that the operator's definition keeps on chip are not counted.
  Roof: arithmetic intensity flops/bytes against the ridge peak_flops / 8 TB/s decides which roof bounds the launch.

DESIGN.md §4 lists the per-unit figures these formulas implement.
"""
from __future__ import annotations

from typing import Optional, Tuple

HBM_PEAK_GBS = 8000.0                             # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}      # dense MFMA peaks at 2.4 GHz


def _elt(dtype_code: int) -> int:
    return 4 if dtype_code == 0 else 2


def _conv_out(H, W, k, s, p):
    return (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1


def model(name: str, a) -> Optional[Tuple[float, float]]:
    """(flops, bytes) of one launch, or None for an entry point with no model (host-side tables, size queries)."""
    P = lambda i: a[i] is not None              # pointer argument present?
    if name == 'rvt_stem_fwd':
        e, F, Cin, h, w, H, W = _elt(a[6]), a[7], a[8], a[10], a[11], a[12], a[13]
        Ho, Wo = _conv_out(H, W, 7, 4, 3)
        return 2.0 * F * Ho * Wo * 64 * Cin * 49, F * Cin * h * w + 2.0 * F * Ho * Wo * 64 * e + 64 * 49 * Cin * e
    if name == 'rvt_stem_wgrad':
        e, F, Cin, h, w, H, W = _elt(a[4]), a[5], a[6], a[8], a[9], a[10], a[11]
        Ho, Wo = _conv_out(H, W, 7, 4, 3)
        return 2.0 * F * Ho * Wo * 64 * Cin * 49, F * Cin * h * w + 1.0 * F * Ho * Wo * 64 * e + 64 * 49 * Cin * 4
    if name == 'rvt_prepack_input':
        src_u8, e, F, Cin, h, w, H, W, Cp = a[1], _elt(a[3]), a[4], a[5], a[6], a[7], a[8], a[9], a[10]
        return 0.0, F * Cin * h * w * (1 if src_u8 else 4) + 1.0 * F * H * W * Cp * e
    if name in ('rvt_conv_fwd', 'rvt_conv_wgrad', 'rvt_conv_dgrad'):
        o = {'rvt_conv_fwd': 3, 'rvt_conv_wgrad': 4, 'rvt_conv_dgrad': 4}[name]
        e, F, H, W, Cin, Cout, k, s, p = _elt(a[o]), *a[o + 1:o + 9]
        Ho, Wo = _conv_out(H, W, k, s, p)
        fl = 2.0 * F * Ho * Wo * Cout * Cin * k * k
        by = 1.0 * F * H * W * Cin * e + 1.0 * F * Ho * Wo * Cout * e + Cout * Cin * k * k * (4 if name.endswith('wgrad') else e)
        if name == 'rvt_conv_dgrad' and P(2):
            by += 1.0 * F * H * W * Cin * e
        return fl, by
    if name == 'rvt_conv_dgrad4':
        e, F, H, W, Cin, Cout = _elt(a[4]), *a[5:10]
        by = 1.0 * F * H * W * Cin * e * (2 if P(2) else 1) + 1.0 * F * (H // 2) * (W // 2) * Cout * e + 9 * Cout * Cin * e
        return 2.0 * F * (H // 2) * (W // 2) * Cout * Cin * 9, by
    if name == 'rvt_layernorm_fwd':
        e, rows, C = _elt(a[4]), a[5], a[6]
        return 0.0, 2.0 * rows * C * e
    if name == 'rvt_layernorm_bwd':
        e, rows, C = _elt(a[7]), a[8], a[9]
        return 0.0, (3.0 + (1 if P(3) else 0)) * rows * C * e
    if name == 'rvt_linear_fwd':
        e, M, N, K = _elt(a[4]), a[5], a[6], a[7]
        return 2.0 * M * N * K, (1.0 * M * (N + K) + N * K) * e
    if name == 'rvt_ln_linear_fwd':                      # u = LN(x), y = u w^T + bias
        e, M, C, N = _elt(a[7]), a[8], a[9], a[10]
        return 2.0 * M * N * C, (1.0 * M * (C * (2 if P(5) and P(1) else 1) + N) + N * C) * e
    if name == 'rvt_linear_gelu_fwd':
        e, M, N, K = _elt(a[5]), a[6], a[7], a[8]
        return 2.0 * M * N * K, (1.0 * M * (N * (2 if P(4) else 1) + K) + N * K) * e
    if name == 'rvt_linear_scale_res_fwd':
        e, M, N, K = _elt(a[6]), a[7], a[8], a[9]
        return 2.0 * M * N * K, (1.0 * M * (2 * N + K) + N * K) * e
    if name == 'rvt_linear_dgrad':                       # dx[M][K] = dy[M][N] wt[K][N]^T (* gelu' | + add | * mul)
        e, M, N, K = _elt(a[6]), a[7], a[8], a[9]
        side = 1 if (P(2) or P(3) or P(4)) else 0
        return 2.0 * M * N * K, (1.0 * M * (N + K * (1 + side)) + N * K) * e
    if name == 'rvt_linear_dgrad_preln':                 # dy0 = LN'(dy w + add; y0): same rows as the form below with add
        e, M, C, K = _elt(a[8]), a[9], a[10], a[11]
        return 2.0 * M * C * K, (1.0 * M * (K + 3 * C) + K * C) * e
    if name == 'rvt_linear_dgrad_ln':                    # dx = add + LN'(dy w; x)
        e, M, C, K = _elt(a[8]), a[9], a[10], a[11]
        return 2.0 * M * C * K, (1.0 * M * (K + C * (2 + (1 if P(3) else 0))) + K * C) * e
    if name == 'rvt_linear_wgrad':
        e, M, N, K = _elt(a[5]), a[6], a[7], a[8]
        return 2.0 * M * N * K, 1.0 * M * (N + K) * e + 4.0 * N * K
    if name == 'rvt_mlp_fwd':                            # LN2 -> fc1 -> GELU -> fc2 -> gamma, + residual
        e, M, C = _elt(a[12]), a[13], a[14]
        rows = 2 + ((8 if P(3) else 4) if P(2) else 0) + (1 if P(4) else 0)        # (g_out alone = the pre-activation only)
        return 16.0 * M * C * C, (1.0 * rows * M * C + 8 * C * C) * e
    if name == 'rvt_mlp_bwd_dgrad':                      # fc2 dgrad * gp -> dh (stored) -> fc1 dgrad -> LN2' + residual
        e, M, C = _elt(a[10]), a[11], a[12]
        return 16.0 * M * C * C, (11.0 * M * C + 8 * C * C) * e
    if name == 'rvt_mlp_bwd_recompute_dgrad':            # both input gradients (16); executed: + recompute of fc1 (8)
        e, M, C = _elt(a[11]), a[12], a[13]
        return 16.0 * M * C * C, (3.0 * M * C + 12 * C * C) * e
    if name == 'rvt_mlp_bwd_recompute_wgrad':            # both weight gradients (16); executed: + recompute of fc1 (8) + fc2 dgrad for dh (8)
        e, M, C = _elt(a[12]), a[13], a[14]
        return 16.0 * M * C * C, 2.0 * M * C * e + 8.0 * C * C * (e + 4)
    if name == 'rvt_mlp_bwd_recompute_both':             # input + weight gradients (32); executed: + recompute of fc1 (8)
        e, M, C = _elt(a[16]), a[17], a[18]
        return 32.0 * M * C * C, 3.0 * M * C * e + 8.0 * C * C * (e + 4)
    if name == 'rvt_attn_fwd':
        e, F, H, W, C, dh, ph, pw = _elt(a[2]), *a[3:10]
        M, L = F * H * W, ph * pw
        return 4.0 * M * L * C, 4.0 * M * C * e
    if name == 'rvt_attn_bwd':                           # dV, dP, dQ, dK (8 L C); executed: + recompute of S (2)
        e, F, H, W, C, dh, ph, pw = _elt(a[3]), *a[4:11]
        M, L = F * H * W, ph * pw
        return 8.0 * M * L * C, 7.0 * M * C * e
    if name == 'rvt_attn_block_fwd':                     # LN1 -> qkv (6C^2) -> attention (4LC) -> proj (2C^2) -> gamma + residual
        e, F, H, W, C, dh, ph, pw = _elt(a[10]), *a[11:18]
        M, L = F * H * W, ph * pw
        return M * (8.0 * C * C + 4.0 * L * C), ((2.0 + (1 if P(2) else 0)) * M * C + 4 * C * C) * e
    if name == 'rvt_attn_block_bwd':                     # dproj (2C^2), core (8LC), du (6C^2); executed: + recompute of qkv (6C^2) + S (2LC)
        e, F, H, W, C, dh, ph, pw = _elt(a[12]), *a[13:20]
        M, L = F * H * W, ph * pw
        return M * (8.0 * C * C + 8.0 * L * C), ((6.0 + (1 if P(4) else 0)) * M * C + 7 * C * C) * e
    if name == 'rvt_attn_block_bwd_preln':               # the same products; rows: x, y0, dxmid in, dy0 + dqkv (3) out
        e, F, H, W, C, dh, ph, pw = _elt(a[11]), *a[12:19]
        M, L = F * H * W, ph * pw
        return M * (8.0 * C * C + 8.0 * L * C), (7.0 * M * C + 7 * C * C) * e
    if name == 'rvt_lstm_fwd':
        e, M, C = _elt(a[8]), a[9], a[10]
        return 16.0 * M * C * C, 1.0 * M * C * (3 * e + 8 + (4 * e if P(7) else 0)) + 8.0 * C * C * e
    if name == 'rvt_lstm_gates_bwd':
        e, M, C = _elt(a[7]), a[8], a[9]
        return 0.0, 1.0 * M * C * ((1 + (1 if P(1) else 0) + 8) * e + 16)
    if name == 'rvt_lstm_dgrad':
        e, M, C = _elt(a[4]), a[5], a[6]
        return 16.0 * M * C * C, (6.0 * M * C + 8 * C * C) * e
    if name == 'rvt_lstm_wgrad':
        e, M, C = _elt(a[6]), a[7], a[8]
        return 16.0 * M * C * C, 6.0 * M * C * e + 32.0 * C * C
    if name == 'rvt_lstm_scan_fwd':                      # per token-step: x in, h out (+ c copy, + gates) — h / c stay on chip
        e, M, C, T = _elt(a[8]), a[9], a[10], a[11]
        rows = 2 + (1 if P(4) else 0) + (4 if P(7) else 0)
        return 16.0 * M * C * C * T, 1.0 * rows * M * C * T * e + 8.0 * M * C + 8 * C * C * e
    if name == 'rvt_lstm_scan_bwd':
        e, M, C, T = _elt(a[17]), a[18], a[19], a[20]
        saved_gates, wgrad, dz = P(16), P(13), P(10)
        fl = 16.0 + (16.0 if wgrad else 0.0)                                 # dgrad (+ in-kernel weight gradient); executed: + gate recompute (16) unless the gates were saved
        rows = (2 + 4 if saved_gates else 4) + 1 + (4 if dz else 0)          # (c, dH, gates | x, h, c, dH) in, dx out (+ dz out)
        return fl * M * C * C * T, 1.0 * rows * M * C * T * e + 16.0 * M * C + 16 * C * C * e
    if name == 'rvt_lstm_scan3_fwd':                     # wide stages: x in, h out (+ bf16 c copy + 4 activated gates when saving); weights once
        e, M, C, T = _elt(a[8]), a[9], a[10], a[11]
        rows = 2 + (5 if P(7) else 0)
        return 16.0 * M * C * C * T, 1.0 * rows * M * C * T * e + 8.0 * M * C + 8 * C * C * e
    if name == 'rvt_lstm_scan3_bwd':                     # gates (4) + c + dH in, dz (4) + dx out; W^T once
        e, M, C, T = _elt(a[10]), a[11], a[12], a[13]
        rows = 4 + 1 + (1 if P(3) else 0) + 4 + 1
        return 16.0 * M * C * C * T, 1.0 * rows * M * C * T * e + 16.0 * M * C + 8 * C * C * e
    if name == 'rvt_lstm_scan3_pack':
        C = a[3]
        return 0.0, 8.0 * C * C * 2 * (1 + (1 if P(1) else 0) + (1 if P(2) else 0))
    if name == 'rvt_dwconv_fwd':
        e, N, H, W, C, k = _elt(a[6]), *a[7:12]
        return 0.0, 2.0 * N * H * W * C * e
    if name == 'rvt_dwconv_wgrad':
        e, N, H, W, C, k = _elt(a[6]), *a[7:12]
        return 0.0, 2.0 * N * H * W * C * e
    if name in ('rvt_token_mask_fwd', 'rvt_token_mask_bwd'):
        e, M, C = _elt(a[3]), a[4], a[5]
        return 0.0, 1.0 * M * C * e
    # ---- detection tail (rows f2 / f3): BatchNorm + SiLU row kernels around the PAFPN / head convolutions, decode, SimOTA + losses
    if name == 'rvt_bn_stats':                                   # one read of the conv output
        e, rows, C = _elt(a[3]), a[4], a[5]
        return 0.0, 1.0 * rows * C * e
    if name == 'rvt_bn_act_fwd':                                 # read + write
        e, rows, C = _elt(a[4]), a[5], a[6]
        return 0.0, 2.0 * rows * C * e
    if name == 'rvt_bn_act_bwd_stats':                           # dy and x in
        e, rows, C = _elt(a[8]), a[9], a[10]
        return 0.0, 2.0 * rows * C * e
    if name == 'rvt_bn_act_bwd_apply':                           # dy and x in, dx out
        e, rows, C = _elt(a[9]), a[10], a[11]
        return 0.0, 3.0 * rows * C * e
    if name == 'rvt_bn_finalize':
        return 0.0, 40.0 * a[13]
    if name in ('rvt_yolox_decode', 'rvt_yolox_decode_bwd'):     # the level's two prediction maps <-> [B][A][5+nc] fp32 rows
        o = 2 if name == 'rvt_yolox_decode' else 5
        ld_ro, ld_cls, e, B, H, W, _st, nc = a[o], a[o + 1], _elt(a[o + 2]), *a[o + 3:o + 8]
        outs = (1 if P(12) else 0) + (1 if P(13) else 0) if name == 'rvt_yolox_decode' else 2
        return 0.0, 1.0 * B * H * W * ((ld_ro + ld_cls) * e + outs * (5 + nc) * 4)
    if name == 'rvt_simota_loss':                                # pred in (cost, select/resolve, loss passes), cost + IoU matrices out and back
        L_, B, G, A, nc = a[4:9]
        return 0.0, 3.0 * B * A * (5 + nc) * 4 + (B * A * (5 + nc) * 4 if P(10) else 0) + 4.0 * B * G * A * 4 + 6.0 * B * A * 4
    return None


def executed(name: str, a) -> Optional[float]:
    """MFMA-side FLOPs the launch actually EXECUTES: model()'s algorithmic figure plus the products a recompute-style backward
    re-does by design.  Equal to the algorithmic figure for every other entry point."""
    m = model(name, a)
    if m is None:
        return None
    fl = m[0]
    P = lambda i: a[i] is not None
    if name == 'rvt_mlp_bwd_recompute_dgrad':
        return fl + 8.0 * a[12] * a[13] * a[13]
    if name == 'rvt_mlp_bwd_recompute_wgrad':
        return fl + 16.0 * a[13] * a[14] * a[14]
    if name == 'rvt_mlp_bwd_recompute_both':
        return fl + 8.0 * a[17] * a[18] * a[18]
    if name == 'rvt_attn_bwd':
        F, H, W, C, dh, ph, pw = a[4:11]
        return fl + 2.0 * F * H * W * ph * pw * C
    if name == 'rvt_attn_block_bwd':
        F, H, W, C, dh, ph, pw = a[13:20]
        return fl + F * H * W * (6.0 * C * C + 2.0 * ph * pw * C)
    if name == 'rvt_attn_block_bwd_preln':
        F, H, W, C, dh, ph, pw = a[12:19]
        return fl + F * H * W * (6.0 * C * C + 2.0 * ph * pw * C)
    if name == 'rvt_lstm_scan_bwd' and not P(16):
        M, C, T = a[18], a[19], a[20]
        return fl + 16.0 * M * C * C * T
    return fl


def roofline_entry(name: str, flops: float, bytes_: float, ms: float, launches: int, dtype: str, mfma_peak_tflops: float = None,
                   executed_flops: float = None):
    """One roofline record for `launches` launches that took `ms` in total.  `flops` = ALGORITHMIC (what `frac` is computed from);
    `executed_flops` (optional) = recompute-inclusive, reported beside it."""
    peak = mfma_peak_tflops or PEAK_TFLOPS[dtype]
    tfl = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    gbs = bytes_ / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    ai = flops / bytes_ if bytes_ > 0 else float('inf')
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    bound = 'hbm' if ai < ridge else 'mfma'
    rec = {'kernel': name, 'bound': bound, 'launches': launches, 'ms': round(ms, 4), 'avg_launch_ms': round(ms / max(launches, 1), 4),
           'algorithmic_gflop': round(flops / 1e9, 2), 'algorithmic_gbyte': round(bytes_ / 1e9, 3),
           'arithmetic_intensity_flop_per_byte': round(ai, 1) if ai != float('inf') else None,
           'hbm_gbs': round(gbs, 1), 'hbm_frac': round(gbs / HBM_PEAK_GBS, 4),
           'mfma_tflops': round(tfl, 2), 'mfma_frac': round(tfl / peak, 4)}
    if executed_flops is not None and executed_flops != flops:
        rec['executed_gflop'] = round(executed_flops / 1e9, 2)
        rec['executed_mfma_tflops'] = round(executed_flops / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0
    if bound == 'hbm':
        rec.update(achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4))
    else:
        rec.update(achieved=round(tfl, 2), peak=peak, unit='TFLOP/s', frac=round(tfl / peak, 4))
    return rec
