"""CPU restatement of the YOLOX PAFPN forward in plain tensor arithmetic.

TEST INFRASTRUCTURE — never imported by rvt_amd; only tests/ compare against it.  Each function cites the reference lines it
follows; it is pinned by fixtures recorded from the unmodified reference (oracle/make_golden_fpn.py -> tests/golden/fpn_*.npz).
Functional over a parameter dict with the reference's state_dict names; BatchNorm buffers are updated in a copy that is returned.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def base_conv(x: Tensor, p: Dict[str, Tensor], pre: str, k: int, stride: int, training: bool, new_stats: Dict[str, Tensor],
              eps: float = 1e-5, momentum: float = 0.1) -> Tensor:
    """Conv2d(bias=False, padding=(k-1)//2) -> BatchNorm2d -> SiLU on an NCHW map.
    reference: yolox/models/network_blocks.py:29-53 (BaseConv), nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1, biased batch
    variance for the normalisation, unbiased for running_var)."""
    w = p[pre + 'conv.weight']
    pad = (k - 1) // 2
    N, Cin, H, W = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    cols = F.unfold(x, kernel_size=k, padding=pad, stride=stride)                   # explicit im2col contraction
    y = torch.einsum('bkn,ok->bon', cols, w.reshape(w.shape[0], -1)).reshape(N, -1, Ho, Wo)
    if training:
        mean = y.mean(dim=(0, 2, 3))
        var = ((y - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        n = y.numel() // y.shape[1]
        new_stats[pre + 'bn.running_mean'] = (1 - momentum) * p[pre + 'bn.running_mean'] + momentum * mean.detach()
        new_stats[pre + 'bn.running_var'] = (1 - momentum) * p[pre + 'bn.running_var'] + momentum * var.detach() * n / max(n - 1, 1)
    else:
        mean, var = p[pre + 'bn.running_mean'], p[pre + 'bn.running_var']
    z = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    z = z * p[pre + 'bn.weight'][None, :, None, None] + p[pre + 'bn.bias'][None, :, None, None]
    return z * torch.sigmoid(z)                                                       # SiLU (network_blocks.py:9-14)


def csp_layer(x: Tensor, p, pre: str, n: int, training: bool, ns) -> Tensor:
    """network_blocks.py:104-141 with shortcut=False (yolo_pafpn.py:57-62): conv1 -> n x (1x1, 3x3) ; conv2 ; cat ; conv3."""
    x1 = base_conv(x, p, pre + 'conv1.', 1, 1, training, ns)
    for i in range(n):
        x1 = base_conv(base_conv(x1, p, f'{pre}m.{i}.conv1.', 1, 1, training, ns), p, f'{pre}m.{i}.conv2.', 3, 1, training, ns)
    x2 = base_conv(x, p, pre + 'conv2.', 1, 1, training, ns)
    return base_conv(torch.cat((x1, x2), 1), p, pre + 'conv3.', 1, 1, training, ns)


def upsample2(x: Tensor) -> Tensor:
    """interpolate(scale_factor=2, mode='nearest-exact') (yolo_pafpn.py:49): out[y][x] = in[y // 2][x // 2]."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def pafpn_forward(feats: Dict[int, Tensor], p: Dict[str, Tensor], depth: float, in_stages=(2, 3, 4), training: bool = False) \
        -> Tuple[Tuple[Tensor, Tensor, Tensor], Dict[str, Tensor]]:
    """yolo_pafpn.py:109-139.  Returns ((pan_out2, pan_out1, pan_out0), updated running statistics)."""
    n = round(3 * depth)
    ns: Dict[str, Tensor] = {}
    x2, x1, x0 = (feats[s] for s in in_stages)
    fpn_out0 = base_conv(x0, p, 'lateral_conv0.', 1, 1, training, ns)
    f_out0 = csp_layer(torch.cat([upsample2(fpn_out0), x1], 1), p, 'C3_p4.', n, training, ns)
    fpn_out1 = base_conv(f_out0, p, 'reduce_conv1.', 1, 1, training, ns)
    pan_out2 = csp_layer(torch.cat([upsample2(fpn_out1), x2], 1), p, 'C3_p3.', n, training, ns)
    p_out1 = base_conv(pan_out2, p, 'bu_conv2.', 3, 2, training, ns)
    pan_out1 = csp_layer(torch.cat([p_out1, fpn_out1], 1), p, 'C3_n3.', n, training, ns)
    p_out0 = base_conv(pan_out1, p, 'bu_conv1.', 3, 2, training, ns)
    pan_out0 = csp_layer(torch.cat([p_out0, fpn_out0], 1), p, 'C3_n4.', n, training, ns)
    return (pan_out2, pan_out1, pan_out0), ns
