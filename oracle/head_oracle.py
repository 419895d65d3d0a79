"""CPU restatement of the YOLOX head tail: decode, SimOTA assignment, detection losses.

TEST INFRASTRUCTURE — never imported by rvt_amd; only tests/ compare against it.  Each function cites the reference lines it
follows; it is pinned by fixtures recorded from the unmodified reference (oracle/make_golden_head.py -> tests/golden/simota_*.npz,
head_*.npz).  Plain torch-CPU tensor arithmetic, one image at a time like the reference; gradients come from autograd.
Ties (equal costs / IoUs) go to the lower index, the rule rvt_amd/csrc/simota.hpp states.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def grids(hws: Sequence[Tuple[int, int]], strides: Sequence[int]):
    """x / y cell index and stride of every anchor, level by level, row-major (yolo_head.py:248-258, :190-197)."""
    xs, ys, st = [], [], []
    for (H, W), s in zip(hws, strides):
        yv, xv = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
        xs.append(xv.reshape(-1).float()); ys.append(yv.reshape(-1).float()); st.append(torch.full((H * W,), float(s)))
    return torch.cat(xs), torch.cat(ys), torch.cat(st)


def decode_train(maps: Sequence[Tensor], hws, strides, nc: int) -> Tensor:
    """Per-level channels-last maps [reg(4)|obj(1)|pad] and [cls(nc)|pad] -> [B][A][5+nc] with decoded boxes and raw logits
    (yolo_head.py:184-189, :259-267)."""
    xs, ys, st = grids(hws, strides)
    rows = []
    for l in range(len(hws)):
        ro, cl = maps[2 * l], maps[2 * l + 1]
        B = ro.shape[0]
        rows.append(torch.cat([ro.reshape(B, -1, ro.shape[-1])[..., :5], cl.reshape(B, -1, cl.shape[-1])[..., :nc]], -1))
    raw = torch.cat(rows, 1).float()
    xy = (raw[..., :2] + torch.stack([xs, ys], -1)) * st[:, None]
    wh = torch.exp(raw[..., 2:4]) * st[:, None]
    return torch.cat([xy, wh, raw[..., 4:]], -1)


def to_infer(pred_train: Tensor) -> Tensor:
    """The returned detections: decoded boxes, sigmoid scores (yolo_head.py:211-214, :269-290)."""
    return torch.cat([pred_train[..., :4], torch.sigmoid(pred_train[..., 4:])], -1)


def pair_iou(a: Tensor, b: Tensor) -> Tensor:
    """utils/boxes.py:79-102 with xyxy=False: a [G][4], b [N][4] in cx cy w h -> [G][N]."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    en = (tl < br).float().prod(2)
    ai = (br - tl).prod(2) * en
    return ai / (a[:, 2:].prod(1)[:, None] + b[:, 2:].prod(1) - ai)


@torch.no_grad()
def assign_image(pred: Tensor, gt: Tensor, xs: Tensor, ys: Tensor, st: Tensor, nc: int):
    """SimOTA for one image (yolo_head.py:453-606).  pred [A][5+nc] decoded, gt [n][5] (class, cx, cy, w, h), n >= 1.
    Returns (match [A] = ground-truth row or -1, matched IoU [A])."""
    A = pred.shape[0]
    xc, yc, rad = (xs + 0.5) * st, (ys + 0.5) * st, 1.5 * st                                     # :550-557
    d = torch.stack([xc - (gt[:, 1:2] - rad), yc - (gt[:, 2:3] - rad), (gt[:, 1:2] + rad) - xc, (gt[:, 2:3] + rad) - yc], 2)
    in_c = d.min(-1).values > 0.0                                                                 # [n][A]  :568-569
    cand = in_c.sum(0) > 0                                                                        # :570
    match = torch.full((A,), -1, dtype=torch.long)
    piou = torch.zeros(A)
    idx = torch.nonzero(cand).reshape(-1)
    if idx.numel() == 0:
        return match, piou
    geom = in_c[:, cand]
    iou = pair_iou(gt[:, 1:5], pred[cand, :4])                                                    # :481
    q = (torch.sigmoid(pred[cand, 5:]) * torch.sigmoid(pred[cand, 4:5])).sqrt()                   # :493-495
    onehot = F.one_hot(gt[:, 0].long(), nc).float()
    cls = -(onehot[:, None, :] * torch.log(q).clamp(min=-100) + (1 - onehot[:, None, :]) * torch.log(1 - q).clamp(min=-100)).sum(-1)
    cost = cls + 3.0 * -torch.log(iou + 1e-8) + 1e6 * (~geom).float()                             # :502-506
    n, m = cost.shape
    M = torch.zeros(n, m, dtype=torch.long)
    top = torch.sort(iou, dim=1, descending=True, stable=True).values[:, :min(10, m)]            # :580-582
    dyn_k = top.sum(1).int().clamp(min=1)
    order = torch.sort(cost, dim=1, stable=True).indices                                          # lowest cost first, ties: lower anchor
    for g in range(n):
        M[g, order[g, :int(dyn_k[g])]] = 1                                                        # :583-586
    multi = M.sum(0) > 1
    if multi.any():                                                                               # :590-594
        amin = torch.min(cost[:, multi], dim=0).indices
        M[:, multi] = 0
        M[amin, multi] = 1
    fg = M.sum(0) > 0
    match[idx[fg]] = M[:, fg].argmax(0)
    piou[idx[fg]] = (M * iou).sum(0)[fg]
    return match, piou


def iou_loss(p: Tensor, t: Tensor) -> Tensor:
    """losses.py:17-35, loss_type 'iou': 1 - iou^2 with iou = I / (U + 1e-16)."""
    tl = torch.max(p[:, :2] - p[:, 2:] / 2, t[:, :2] - t[:, 2:] / 2)
    br = torch.min(p[:, :2] + p[:, 2:] / 2, t[:, :2] + t[:, 2:] / 2)
    en = (tl < br).float().prod(1)
    ai = (br - tl).prod(1) * en
    return 1 - (ai / (p[:, 2:].prod(1) + t[:, 2:].prod(1) - ai + 1e-16)) ** 2


def head_losses(pred_train: Tensor, labels: Tensor, hws, strides, nc: int):
    """yolo_head.py:291-443.  Returns (losses [5] = loss, 5 iou, obj, cls, num_fg / max(num_gts, 1); match [B][A]; matched IoU)."""
    xs, ys, st = grids(hws, strides)
    B, A, _ = pred_train.shape
    nlabel = (labels.sum(2) > 0).sum(1)                                                          # :309
    matches, pious = [], []
    for b in range(B):
        n = int(nlabel[b])
        if n == 0:
            matches.append(torch.full((A,), -1, dtype=torch.long)); pious.append(torch.zeros(A))
            continue
        mt, pi = assign_image(pred_train[b].detach(), labels[b, :n].float(), xs, ys, st, nc)
        matches.append(mt); pious.append(pi)
    match, piou = torch.stack(matches), torch.stack(pious)
    fg = match >= 0
    num_fg = max(int(fg.sum()), 1)
    bi = torch.nonzero(fg)[:, 0]
    tgt = labels[bi, match[fg]].float()                                                           # [nfg][5]
    l_iou = iou_loss(pred_train[fg][:, :4], tgt[:, 1:5]).sum() / num_fg                          # :414-416
    l_obj = F.binary_cross_entropy_with_logits(pred_train[..., 4], fg.float(), reduction='sum') / num_fg
    cls_t = F.one_hot(tgt[:, 0].long(), nc).float() * piou[fg][:, None]                           # :386-388
    l_cls = F.binary_cross_entropy_with_logits(pred_train[fg][:, 5:], cls_t, reduction='sum') / num_fg
    ratio = torch.tensor(num_fg / max(int(nlabel.sum()), 1))
    return torch.stack([5.0 * l_iou + l_obj + l_cls, 5.0 * l_iou, l_obj, l_cls, ratio]), match, piou
