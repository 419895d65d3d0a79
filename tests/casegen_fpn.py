"""Deterministic PAFPN test cases: parameters, inputs and cotangents from numpy seeds, so that the
golden fixtures (recorded from the reference by oracle/make_golden_fpn.py) need to store outputs only."""
import zlib

import numpy as np

CASES = {
    # channels of stages 2-4, depth (round(3 * depth) bottlenecks per CSP layer), batch, stage-2 resolution (stage 3 / 4 = /2, /4)
    'fpn_micro': dict(in_channels=(32, 64, 128), depth=0.34, N=2, hw2=(8, 12)),
    'fpn_base': dict(in_channels=(128, 256, 512), depth=0.67, N=2, hw2=(12, 20)),       # RVT-Base widths, config/model/maxvit_yolox/default.yaml:44-52
}


def _rng(name: str, what: str):
    return np.random.default_rng(zlib.crc32(f'{name}/{what}'.encode()))


def make_params(name: str, shapes):
    """shapes: [(state_dict name, shape)] in module order -> dict of float32 arrays (num_batches_tracked: int64 zeros)."""
    out = {}
    for k, shp in shapes:
        r = _rng(name, k)
        if k.endswith('num_batches_tracked'):
            out[k] = np.zeros(shp, dtype=np.int64)
        elif k.endswith('conv.weight'):
            fan_in = shp[1] * shp[2] * shp[3]
            out[k] = (r.standard_normal(shp) / np.sqrt(fan_in)).astype(np.float32)
        elif k.endswith('bn.weight') or k.endswith('running_var'):
            out[k] = r.uniform(0.5, 1.5, shp).astype(np.float32)
        else:                                   # bn.bias, running_mean
            out[k] = (0.1 * r.standard_normal(shp)).astype(np.float32)
    return out


def make_inputs(name: str):
    c = CASES[name]
    H, W = c['hw2']
    return {s: _rng(name, f'x{s}').standard_normal((c['N'], ch, H >> i, W >> i)).astype(np.float32)
            for i, (s, ch) in enumerate(zip((2, 3, 4), c['in_channels']))}


def make_cotangents(name: str):
    c = CASES[name]
    H, W = c['hw2']
    return [_rng(name, f'cot{i}').standard_normal((c['N'], ch, H >> i, W >> i)).astype(np.float32) for i, ch in enumerate(c['in_channels'])]
