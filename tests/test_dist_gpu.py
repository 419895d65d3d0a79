"""RCCL code path on the single gpurun GPU: backend "nccl" at world_size 1 executes the same collectives (bucketed
all-reduce launched from the stage-major backward, AVG in the collective) that the driver's N>1 bench uses.  GPU only."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from rvt_amd import RNNDetector, backbone_config
from rvt_amd.dist import StageGradReducer
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
def grads(reducer):
    torch.manual_seed(0)
    m = RNNDetector(backbone_config('tiny', 'gen1'), compute_dtype=torch.bfloat16).to(dev)
    if reducer is not None:
        reducer.attach(m)
    g = torch.Generator(device=dev).manual_seed(1)
    xs = torch.randint(0, 11, (2, 2, 20, 240, 304), generator=g, dtype=torch.uint8, device=dev)
    feats, _ = m.forward_sequence(xs)
    torch.autograd.backward([feats[s] for s in (2, 3, 4)], [torch.ones_like(feats[s]) for s in (2, 3, 4)])
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters()}
red = StageGradReducer(force=True)           # run the RCCL all-reduce even though world_size == 1
a = grads(red)
b = grads(None)
for n in a:                                   # mean over one rank == identity (up to the order of the fp32 atomics that
    err = float((a[n] - b[n]).abs().max()) / max(float(b[n].abs().max()), 1e-30)      # fold the LayerNorm gradients)
    assert err < 1e-4, (n, err)
assert len(red._pending) == 0
dist.destroy_process_group()
print('NCCL_OK')
'''


def test_rccl_bucket_allreduce_world1(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29741', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'NCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_under_torchrun_world1():
    """bench.py exactly as the driver launches it for N>1, with one rank."""
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                        '--master-addr', '127.0.0.1', '--master-port', '29743', os.path.join(ROOT, 'bench.py'),
                        '--gpus', '1', '--steps', '2', '--warmup', '1', '--workload', 'tiny_gen1', '--no-cpu-baseline',
                        '--force-reducer'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert '"value"' in r.stdout


def test_bench_gpus2_self_launch_reports_missing_device():
    """`python bench.py --gpus 2` (no launcher) starts two ranks itself; on a one-GPU box rank 1 says which device is missing."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('>= 2 GPUs here: the launch would run the real 2-GPU bench')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--workload', 'tiny_gen1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0
    assert 'launching 2 ranks under torch.distributed.run' in r.stderr
    assert 'needs cuda:1 but only 1 device(s) are visible' in r.stderr, r.stderr[-3000:]


def test_bench_default_line_carries_also_configs():
    """The N=1 headline line carries BASELINE configs[1] and configs[4] under `also` (driver-visible)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    also = line['also']
    assert also['tiny_gen1']['value'] > 0 and also['tiny_gen1']['ms_per_step'] > 0, also
    assert also['stream_latency']['p50'] > 0 and also['stream_latency']['p99'] >= also['stream_latency']['p50'], also
    assert line['roofline']['frac'] < 1 and line['n_gpus'] == 1
