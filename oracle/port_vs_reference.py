"""Time the oracle PORT (oracle/rvt_oracle.py) and the UNMODIFIED reference backbone on the same CPU, same inputs.

TEST / MEASUREMENT INFRASTRUCTURE.  Runs only in the authoring container (/root/reference does not exist on the GPU
box).  bench.py's `cpu_baseline` leg times the port on the GPU box's host cores (`kind: "port"`); this script measures
how the port compares with the real reference so that the bench line can state the ratio with provenance
(the port must not silently understate the reference).

    python oracle/port_vs_reference.py [B] [T]      ->  one JSON line (commit it as profiles/r4/port_vs_reference.json)

Workload: RVT-Base, 1 Mpx (20x360x640 uint8, padded to 384x640), fp32, forward + backward of
sum(stage 2-4 features over all T), all usable cores, best of 3 — the probe shape of BASELINE.md §3 (B = 2, T = 3).
"""
from __future__ import annotations

import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, '_stubs'))
sys.path.insert(1, '/root/reference')
sys.path.insert(2, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tests.cores import usable_cores  # noqa: E402


def best_of(fn, n=3):
    best = float('inf')
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    from oracle import rvt_oracle as O
    from oracle.make_golden import reference_cfg
    from rvt_amd import backbone_config
    cfgd = backbone_config('base', 'gen4')
    ref_cfg = reference_cfg(dict(input_channels=20, enable_masking=False, embed_dim=64, dim_multiplier=(1, 2, 4, 8),
                                 num_blocks=(1, 1, 1, 1), patch_size=4, overlap=True, partition_size=(6, 10), dim_head=32,
                                 dws_conv=False, dws_conv_only_hidden=True, dws_conv_kernel_size=3))
    from models.detection.recurrent_backbone import build_recurrent_backbone
    torch.manual_seed(0)
    ref = build_recurrent_backbone(ref_cfg)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in ref.state_dict().items()}
    ocfg = O.OracleCfg(embed_dim=64, dim_head=32, partition_size=(6, 10), conv_impl='aten')      # as bench.py's cpu_baseline leg
    g = torch.Generator().manual_seed(1)
    xs = torch.randint(0, 11, (T, B, 20, 360, 640), generator=g, dtype=torch.uint8)
    Hm, Wm = cfgd.in_res_hw

    def run_reference():
        ref.zero_grad(set_to_none=True)
        states, loss = None, 0.0
        for t in range(T):                       # the time loop of modules/detection.py:131-148
            x = F.pad(xs[t].float(), [0, Wm - 640, 0, Hm - 360])
            out, states = ref(x, states)
            loss = loss + sum(out[s].sum() for s in (2, 3, 4))
        loss.backward()

    def run_port():
        feats, _ = O.sequence_forward(xs, None, params, ocfg, (Hm, Wm))
        loss = sum(feats[t][s].sum() for t in range(T) for s in (2, 3, 4))
        torch.autograd.grad(loss, list(params.values()), allow_unused=True)

    t_ref = t_port = float('inf')
    for _ in range(3):                           # interleaved: the container's vCPUs are shared and noisy
        t_ref = min(t_ref, best_of(run_reference, 2))
        t_port = min(t_port, best_of(run_port, 2))
    print(json.dumps(dict(workload=f'RVT-Base 1Mpx fp32 fwd+bwd, B={B}, T={T}', cores=ncores, torch=torch.__version__,
                          reference_s=round(t_ref, 3), reference_event_tensors_per_s=round(B * T / t_ref, 2),
                          port_s=round(t_port, 3), port_event_tensors_per_s=round(B * T / t_port, 2),
                          port_over_reference=round(t_ref / t_port, 3))))


if __name__ == '__main__':
    main()
