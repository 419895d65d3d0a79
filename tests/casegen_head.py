"""Deterministic YOLOX-head / SimOTA test cases: parameters, feature maps, prediction maps and labels
from numpy seeds, so that the golden fixtures (recorded from the reference by oracle/make_golden_head.py) store outputs only."""
import zlib

import numpy as np

# whole head: FPN maps -> towers -> predictions -> decode / assignment / losses
CASES = {
    # channels of the FPN maps (hidden = in_channels[-1] / 4), classes, images, level-0 resolution (levels 1 / 2 = /2, /4), labels
    'head_micro': dict(in_channels=(32, 64, 128), nc=3, N=3, hw0=(12, 20), strides=(8, 16, 32), G=6, n_gt=(4, 0, 6)),
    'head_base': dict(in_channels=(128, 256, 512), nc=2, N=2, hw0=(12, 20), strides=(8, 16, 32), G=5, n_gt=(5, 2)),   # RVT-Base widths, Gen1 classes
}
# the tail alone on crafted prediction maps (decode + SimOTA + losses): dense scenes, conflicts, the full 1 Mpx anchor count
SIMOTA_CASES = {
    'simota_small': dict(nc=3, N=3, hws=((6, 10), (3, 5), (2, 3)), strides=(8, 16, 32), G=5, n_gt=(3, 0, 5), crowd=False),
    'simota_crowd': dict(nc=2, N=2, hws=((24, 40), (12, 20), (6, 10)), strides=(8, 16, 32), G=12, n_gt=(12, 9), crowd=True),
    'simota_1mpx': dict(nc=3, N=4, hws=((48, 80), (24, 40), (12, 20)), strides=(8, 16, 32), G=20, n_gt=(20, 7, 0, 13), crowd=True),  # A = 5040 (384x640)
}


def _rng(name: str, what: str):
    return np.random.default_rng(zlib.crc32(f'{name}/{what}'.encode()))


def level_hws(c):
    if 'hws' in c:
        return [tuple(h) for h in c['hws']]
    H, W = c['hw0']
    return [(H >> i, W >> i) for i in range(3)]


def image_hw(c):
    (H, W), s = level_hws(c)[0], c['strides'][0]
    return H * s, W * s


def make_labels(name: str, c) -> np.ndarray:
    """[N][G][5] rows (class, cx, cy, w, h); the real rows first, zero rows pad (data/genx_utils/labels.py yolox format)."""
    r = _rng(name, 'labels')
    Hi, Wi = image_hw(c)
    lab = np.zeros((c['N'], c['G'], 5), dtype=np.float32)
    for b, n in enumerate(c['n_gt']):
        anchor = r.uniform([0.2 * Wi, 0.2 * Hi], [0.8 * Wi, 0.8 * Hi])
        for g in range(n):
            w, h = r.uniform(0.08, 0.45) * Wi, r.uniform(0.1, 0.5) * Hi
            if c.get('crowd') and g % 3:                                       # clustered boxes: shared candidate anchors, conflicts
                cx, cy = anchor + r.normal(0, 0.06, 2) * [Wi, Hi]
            else:
                cx, cy = r.uniform(0.02 * Wi, 0.98 * Wi), r.uniform(0.02 * Hi, 0.98 * Hi)   # includes boxes hugging the border
            cx, cy = float(np.clip(cx, 1.0, Wi - 1.0)), float(np.clip(cy, 1.0, Hi - 1.0))
            lab[b, g] = [float(r.integers(0, c['nc'])), cx, cy, w, h]
    return lab


def make_params(name: str, shapes):
    out = {}
    for k, shp in shapes:
        r = _rng(name, k)
        if k.endswith('num_batches_tracked'):
            out[k] = np.zeros(shp, dtype=np.int64)
        elif k.endswith('conv.weight') or (k.endswith('.weight') and 'preds' in k):
            out[k] = (r.standard_normal(shp) / np.sqrt(shp[1] * shp[2] * shp[3])).astype(np.float32)
        elif 'preds' in k and k.endswith('.bias'):
            prior = 0.0 if k.startswith('reg_preds') else -2.0               # (a milder prior than 0.01 so that scores are not all ~0)
            out[k] = (prior + 0.3 * r.standard_normal(shp)).astype(np.float32)
        elif k.endswith('bn.weight') or k.endswith('running_var'):
            out[k] = r.uniform(0.5, 1.5, shp).astype(np.float32)
        else:
            out[k] = (0.1 * r.standard_normal(shp)).astype(np.float32)
    return out


def make_inputs(name: str):
    c = CASES[name]
    return [_rng(name, f'x{i}').standard_normal((c['N'], ch, h, w)).astype(np.float32)
            for i, (ch, (h, w)) in enumerate(zip(c['in_channels'], level_hws(c)))]


def make_pred_maps(name: str):
    """Crafted per-level prediction maps for the SIMOTA_CASES: [reg(4) | obj(1) | pad(3)] and [cls(nc) | pad] channels-last,
    plus the labels.  Anchors near a ground truth regress towards it (with noise) so that IoUs - hence the dynamic k - are large."""
    c = SIMOTA_CASES[name]
    lab = make_labels(name, c)
    r = _rng(name, 'maps')
    nc, N = c['nc'], c['N']
    ncp = (nc + 7) // 8 * 8
    maps = []
    for (H, W), s in zip(level_hws(c), c['strides']):
        ro = np.zeros((N, H, W, 8), dtype=np.float32)
        cl = np.zeros((N, H, W, ncp), dtype=np.float32)
        ro[..., :2] = r.normal(0, 0.5, (N, H, W, 2))
        ro[..., 2:4] = r.normal(0.5, 0.6, (N, H, W, 2))
        ro[..., 4] = r.normal(-1.0, 2.0, (N, H, W))
        cl[..., :nc] = r.normal(-0.5, 2.0, (N, H, W, nc))
        gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
        for b in range(N):
            for g in range(c['n_gt'][b]):
                k, cx, cy, w, h = lab[b, g]
                near = (np.abs((gx + 0.5) * s - cx) < 2.5 * s) & (np.abs((gy + 0.5) * s - cy) < 2.5 * s) & (r.random((H, W)) < 0.7)
                n = int(near.sum())
                ro[b, near, 0] = cx / s - gx[near] + r.normal(0, 0.35, n)
                ro[b, near, 1] = cy / s - gy[near] + r.normal(0, 0.35, n)
                ro[b, near, 2] = np.log(w / s) + r.normal(0, 0.25, n)
                ro[b, near, 3] = np.log(h / s) + r.normal(0, 0.25, n)
                ro[b, near, 4] = r.normal(1.0, 1.5, n)
                cl[b, near, int(k)] = r.normal(1.0, 1.5, n)
        maps += [ro, cl]
    return maps, lab
