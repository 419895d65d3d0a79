#!/usr/bin/env python3
"""Benchmark of the hot path: RVT backbone forward + BPTT backward (+ optimizer step, + gradient
all-reduce when N>1) over one synthetic event-tensor sequence batch.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload base_1mpx|tiny_gen1] [--dtype bf16|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: without a launcher (`WORLD_SIZE` unset) the first form starts its N ranks itself under
torch.distributed.run (`self_launch`).  The default N=1 line also carries `also`: BASELINE configs[1] (RVT-Tiny / Gen1
training step) and configs[4] (streaming-inference latency) measured right after the headline run.

Workload (default): BASELINE.json configs[2] — RVT-Base, 1 Mpx shape (20x360x640 uint8, padded to 384x640
by the model), T=21, batch=24 per GPU, bf16, random-init weights, inputs resident in HBM.
A "step" = forward_sequence over (T,B) + backward with random upstream gradients on the stage 2/3/4
features of all T frames (what the FPN consumes) + fused AdamW on the backbone parameters; with N>1 the
per-stage gradient buckets are all-reduced over RCCL, overlapped with the remaining backward.
Metric: event-tensors/s = N*B*T / step time (weak scaling: per-GPU work fixed).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import opmodel  # noqa: E402  (algorithmic FLOP / byte model of every C-ABI entry point)

# ALGORITHMIC matmul FLOPs per event-tensor (SURVEY.md §8d / BASELINE.md §2; 2*MAC of conv+linear+attention,
# fwd+bwd = 3*fwd - stem dgrad)
WORKLOADS = {
    'base_1mpx': dict(size='base', dataset='gen4', hw=(360, 640), T=21, B=24, f_fwd=20.616e9, f_fwdbwd=59.922e9,
                      label='RVT-Base, 1Mpx 20x360x640 (padded 384x640), T=21, B=24/GPU'),
    'tiny_gen1': dict(size='tiny', dataset='gen1', hw=(240, 304), T=21, B=8, f_fwd=2.001e9, f_fwdbwd=5.683e9,
                      label='RVT-Tiny, Gen1 20x240x304 (padded 256x320), T=21, B=8/GPU'),
}
PEAK_TFLOPS = opmodel.PEAK_TFLOPS                 # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
HBM_PEAK_GBS = opmodel.HBM_PEAK_GBS               # MI355X HBM3E (MI355X_MICROARCH.md)


class _Ms:
    """A measured duration standing in for a (start, end) event pair: `a.elapsed_time(b)` of the records below."""

    def __init__(self, ms):
        self.ms = ms

    def elapsed_time(self, _other):
        return self.ms


class OpTimer:
    """HIP-event timing of individual C-ABI launches on torch's current stream (the stream the kernels are
    launched on).  Used for the `roofline` objects: per-launch duration of every entry point.

    The events come from a pool created (and recorded once) BEFORE the instrumented step: creating ~1400 events inside a step of
    ~700 launches made the HIP runtime refill its signal pool in the middle of it, and whichever launch sat behind that refill was
    charged tens of milliseconds (RVT-Tiny: one `rvt_lstm_dgrad` of 18 us measured 50 ms and became the "dominant kernel").
    `two_pass_min` additionally runs the instrumented step twice and keeps the smaller duration of every launch."""

    def __init__(self):
        self.records = {}
        self.enabled_for = None     # None = all ops, or a set of names
        self.pool = []

    def preallocate(self, n):
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for e in self.pool:
            e.record()
        torch.cuda.synchronize()

    def _event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def install(self):
        from rvt_amd import _lib
        self._orig = _lib.call
        timer = self

        def timed_call(name, *args):
            if timer.enabled_for is not None and name not in timer.enabled_for:
                return timer._orig(name, *args)
            e0, e1 = timer._event(), timer._event()
            e0.record()
            timer._orig(name, *args)
            e1.record()
            timer.records.setdefault(name, []).append((e0, e1, args))
        _lib.call = timed_call
        # ops.py binds `L.call` at call time through the module attribute, so patching _lib.call suffices

    def uninstall(self):
        from rvt_amd import _lib
        _lib.call = self._orig

    def freeze(self):
        """Replace the event pairs of the recorded launches by their durations (events go back to the pool)."""
        torch.cuda.synchronize()
        for name, recs in self.records.items():
            for i, (a, b, ar) in enumerate(recs):
                if not isinstance(a, _Ms):
                    recs[i] = (_Ms(a.elapsed_time(b)), None, ar)
                    self.pool += [a, b]

    def two_pass_min(self, step):
        """One more instrumented step; every launch keeps the smaller of its two durations (same launch sequence both times)."""
        self.freeze()
        first, self.records = self.records, {}
        step()
        self.freeze()
        if {k: len(v) for k, v in first.items()} == {k: len(v) for k, v in self.records.items()}:
            for name, recs in self.records.items():
                for i, (a, _, ar) in enumerate(recs):
                    recs[i] = (_Ms(min(a.ms, first[name][i][0].ms)), None, ar)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[name] = dict(calls=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms), max_ms=max(ms))
        return out


def shape_key(args):
    """The integer (non-pointer-sized) arguments of a launch: dtype and the problem sizes - what tells two kernels of one
    entry point apart (pointers are > 2^31 or None)."""
    return tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool) and 0 <= x < (1 << 31))[-8:]


def kernel_groups(records):
    """(entry point, shape key) -> [launch records]: one group = one kernel at one problem shape."""
    groups = {}
    for n, recs in records.items():
        for r in recs:
            groups.setdefault((n, shape_key(r[2])), []).append(r)
    return groups


def group_roofline(key, recs, dtype_name, mfma_peak=None):
    n, shp = key
    ms = sum(a.elapsed_time(b) for a, b, _ in recs)
    fb = [opmodel.model(n, r[2]) for r in recs]
    if any(x is None for x in fb):
        return None
    rec = opmodel.roofline_entry(n, sum(x[0] for x in fb), sum(x[1] for x in fb), ms, len(recs), dtype_name, mfma_peak,
                                 executed_flops=sum(opmodel.executed(n, r[2]) for r in recs))
    rec['shape'] = list(shp)
    return rec


# kernel instantiation behind an (entry point, shape) group, as substrings of the mangled name rocprofv3 prints - used to
# look the group's measured HBM traffic up in profiles/latest_traffic.json.  Only groups that map to ONE instantiation
# launched at ONE shape per step are listed (others share a kernel between shapes: their per-launch average would not
# belong to the group, and the bench reports null rather than a number that does not).
def traffic_lookup(workload_key, key):
    n, shp = key
    subs = None
    if n == 'rvt_lstm_scan_bwd':
        subs = ['lstm_scan_bwd_kernel', f'DF16bLi{shp[-3]}E']
    elif n == 'rvt_lstm_scan_fwd':
        subs = ['lstm_scan_fwd_kernel', f'DF16bLi{shp[-3]}E']
    elif n == 'rvt_mlp_fwd':
        subs = ['mlpc_fwd_kernel' if shp[2] == 64 else 'mlps_fwd_kernel', f'DF16bLi{shp[2]}E']      # (dtype, M, C[, stream 0])
    elif n == 'rvt_mlp_bwd_recompute_dgrad':
        subs = ['mlpc_bwd_dgrad_kernel' if shp[2] == 64 else 'mlps_bwd_dgrad_kernel', 'DF16b']
    elif n in ('rvt_mlp_bwd_recompute_wgrad', 'rvt_mlp_bwd_recompute_both'):
        subs = ['mlpc_bwd_wgrad_kernel' if shp[2] == 64 else 'mlps_bwd_wgrad_kernel']
    elif n == 'rvt_ln_linear_fwd':
        subs = ['lnlin_fwd_kernel']
    elif n == 'rvt_stem_fwd':
        subs = ['stem_fwd_kernel']
    elif n == 'rvt_stem_wgrad':
        subs = ['stem_wgrad_kernel']
    elif n in ('rvt_attn_block_bwd', 'rvt_attn_block_fwd'):
        subs = [n[4:] + '_kernel', 'Li2ELb0ELi' if shp[-1] == 1 else 'Li2ELb1ELi']      # (RVT-Base: the window block of a stage has no norm1, the grid block has)
    elif n == 'rvt_attn_block_bwd_preln':
        subs = ['attn_block_bwd_kernel', 'Li2ELb0ELi4ELb1E']                            # (LN = false, four waves, PRE = true)
    if subs is None:
        return None
    try:
        with open(os.path.join(ROOT, 'profiles', 'latest_traffic.json')) as f:
            t = json.load(f)[workload_key]['kernels']
        hits = [v for k, v in t.items() if all(x in k for x in subs)]
        return hits[0]['traffic_bytes_per_launch'] if len(hits) == 1 else None
    except Exception:
        return None


def step_traffic(workload_key):
    try:
        with open(os.path.join(ROOT, 'profiles', 'latest_traffic.json')) as f:
            return json.load(f)[workload_key]['traffic_bytes_per_step']
    except Exception:
        return None


def mfma_peak_sustained(device):
    """bf16 MFMA rate this part sustains on a pure-MFMA kernel at the clock it actually runs (rvt_probe_mfma), TFLOP/s."""
    from rvt_amd import _lib
    lib = _lib.get_lib()
    scratch = torch.empty(4096 * 256, dtype=torch.float32, device=device)
    st = _lib.stream_of(scratch)
    best = 0.0
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.rvt_probe_mfma(scratch.data_ptr(), 20000, 4096, st)
        e1.record()
        torch.cuda.synchronize()
        if it:
            best = max(best, flops / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def build_model(wl, dtype, device):
    from rvt_amd import RNNDetector, backbone_config
    torch.manual_seed(0)
    cfg = backbone_config(wl['size'], wl['dataset'])
    m = RNNDetector(cfg, compute_dtype=dtype).to(device)
    m.zero_copy_grads = True      # .grad = persistent views of the stage buckets (stable addresses: fused optimizer, hipGraph, RCCL in place)
    return m


def make_batch(wl, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)      # dense random 0..10 (DVFS-honest: not zero-heavy)
    return torch.randint(0, 11, (wl['T'], wl['B'], 20, *wl['hw']), generator=g, dtype=torch.uint8, device=device)


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota (a container
    can see every core of the box yet be throttled to a few; oversubscribing OpenMP there is pathological)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline_worker(workload: str):
    """Runs in a child process (bounded by a timeout in the parent): the CPU oracle — a port of the
    reference algorithm, oracle/rvt_oracle.py — on a bounded sample of the same workload."""
    from oracle import rvt_oracle as O
    from rvt_amd import backbone_config
    wl = WORKLOADS[workload]
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    cfgd = backbone_config(wl['size'], wl['dataset'])
    cfg = O.OracleCfg(embed_dim=cfgd.embed_dim, dim_head=cfgd.stage.attention.dim_head,
                      partition_size=tuple(cfgd.stage.attention.partition_size), conv_impl='aten')
    m = build_model(wl, torch.float32, 'cpu')
    params = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    B, T = 2, 3                    # the probe shape of BASELINE.md §3 (reference measured there: 6.2 event-tensors/s on 8 vCPUs)
    g = torch.Generator().manual_seed(1)
    xs = torch.randint(0, 11, (T, B, 20, *wl['hw']), generator=g, dtype=torch.uint8)
    best, t_start, passes = float('inf'), time.perf_counter(), 0
    while passes < 5 and (passes == 0 or time.perf_counter() - t_start < 25.0):
        t0 = time.perf_counter()
        feats, _ = O.sequence_forward(xs, None, params, cfg, tuple(cfgd.in_res_hw))
        loss = sum(feats[t][s].sum() for t in range(T) for s in (2, 3, 4))
        torch.autograd.grad(loss, list(params.values()), allow_unused=True)
        best = min(best, time.perf_counter() - t0)
        passes += 1
    ratio = None
    try:        # port / reference speed ratio measured in the authoring container (oracle/port_vs_reference.py; same ops, same cores)
        with open(os.path.join(ROOT, 'profiles', 'r4', 'port_vs_reference.json')) as f:
            ratio = json.load(f)['port_over_reference']
    except Exception:
        pass
    out = dict(value=round(B * T / best, 3), unit='event-tensors/s', cores=ncores, kind='port', estimate=True,
               sample=f'{wl["label"].split(",")[0]} at the same resolution, B={B}, T={T}, fp32, fwd+bwd, best of {passes} '
                      f'({best:.2f} s per pass), {ncores} torch CPU threads; oracle in its ATen-op timing mode '
                      f'(F.conv2d / F.layer_norm / F.gelu, as the reference calls them)')
    if ratio and workload == 'base_1mpx':
        out['port_over_reference'] = ratio
        out['reference_estimate'] = round(B * T / best / ratio, 3)
        out['reference_provenance'] = ('the unmodified reference cannot run on the GPU box; in the authoring container (8 vCPUs) the same '
                                       'B=2, T=3 probe ran the reference and the port side by side: profiles/r4/port_vs_reference.json; '
                                       'survey-time reference probe: 6.2 event-tensors/s on 8 vCPUs (BASELINE.md section 3)')
    # BASELINE.json configs[0] as well (RVT-Tiny, Gen1 shape, T=5, batch=2, CPU-only forward - the reference's own plumbing case)
    try:
        wl0 = WORKLOADS['tiny_gen1']
        cfg0d = backbone_config(wl0['size'], wl0['dataset'])
        cfg0 = O.OracleCfg(embed_dim=cfg0d.embed_dim, dim_head=cfg0d.stage.attention.dim_head,
                           partition_size=tuple(cfg0d.stage.attention.partition_size), conv_impl='aten')
        m0 = build_model(wl0, torch.float32, 'cpu')
        p0 = {k: v.detach().clone() for k, v in m0.state_dict().items()}
        x0 = torch.randint(0, 11, (5, 2, 20, *wl0['hw']), generator=g, dtype=torch.uint8)
        b0 = float('inf')
        with torch.no_grad():
            for _ in range(3):
                t0 = time.perf_counter()
                O.sequence_forward(x0, None, p0, cfg0, tuple(cfg0d.in_res_hw))
                b0 = min(b0, time.perf_counter() - t0)
        out['configs0'] = dict(value=round(10 / b0, 2), unit='event-tensors/s', kind='port', estimate=True,
                               sample=f'BASELINE configs[0]: RVT-Tiny, Gen1 20x240x304, T=5, batch=2, fp32 forward only, best of 3 ({b0:.3f} s), '
                                      f'{ncores} torch CPU threads (survey-time probe of the unmodified reference: 84.6 event-tensors/s on 8 vCPUs)')
    except Exception as e:
        out['configs0'] = dict(value=None, note=f'{type(e).__name__}: {e}'[:200])
    print(json.dumps(out))


def cpu_baseline(workload: str):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--workload', workload],
                           capture_output=True, text=True, timeout=150, env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # never let the baseline leg sink the GPU measurement
        return dict(value=None, unit='event-tensors/s', cores=usable_cores(), kind='port',
                    sample=f'cpu baseline did not finish within its 150 s bound ({type(e).__name__})')


def stream_latency(wl, dtype, device, args, with_detector=True):
    """BASELINE configs[4]: one forward step (T=1) on a batch of 64 streams with the ConvLSTM state carried across
    steps (the validation / deployment path, modules/detection.py:231-255), latency per step.  Returns the record."""
    model = build_model(wl, dtype, device)
    Bs = args.batch or 64
    g = torch.Generator(device=device).manual_seed(3)
    frames = [torch.randint(0, 11, (Bs, 20, *wl['hw']), generator=g, dtype=torch.uint8, device=device) for _ in range(4)]
    states = None
    lat = []
    import gc
    with torch.no_grad():
        for i in range(args.warmup + 8):                       # pre-warm 8 steps
            _, states = model(frames[i % 4], states)
        torch.cuda.synchronize()
        # a generation-2 collection over the module / tensor object graph lands inside one step in a few dozen and takes 5 - 10 ms
        # (round 4: wall p99 14.6 ms against GPU p99 4.7 ms): collect now, freeze the survivors, and keep the collector out of the
        # latency path, as a serving loop would (it runs it between requests)
        gc.collect()
        gc.freeze()
        gc.disable()
        for i in range(max(args.steps, 50)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            _, states = model(frames[i % 4], states)
            e1.record()
            t_host = time.perf_counter() - t0                  # the call has returned: everything is enqueued
            torch.cuda.synchronize()
            lat.append((1e3 * (time.perf_counter() - t0), e0.elapsed_time(e1), 1e3 * t_host))
    gc.enable()
    wall = sorted(x[0] for x in lat)
    gpu = sorted(x[1] for x in lat)
    host = sorted(x[2] for x in lat)
    pct = lambda a, q: a[min(len(a) - 1, int(q * len(a)))]
    rec = {'metric': 'streaming-inference step latency (T=1, persistent ConvLSTM state)', 'unit': 'ms',
           'batch': Bs, 'dtype': args.dtype, 'config': {'workload': wl['label'].split(',')[0] + f', B={Bs}, T=1'},
           'wall_p50': round(pct(wall, 0.5), 3), 'wall_p99': round(pct(wall, 0.99), 3),
           'gpu_p50': round(pct(gpu, 0.5), 3), 'gpu_p99': round(pct(gpu, 0.99), 3),
           'host_enqueue_p50': round(pct(host, 0.5), 3),
           'mfma_frac_p50': round(Bs * wl['f_fwd'] / (pct(wall, 0.5) * 1e-3) / (PEAK_TFLOPS[args.dtype] * 1e12), 4),
           'event_tensors_per_s_p50': round(Bs / pct(wall, 0.5) * 1e3, 1),
           'steps_timed': len(lat), 'higher_is_better': False, 'data': 'synthetic'}
    # the same step replayed as ONE hipGraph (rvt_amd.graph.GraphedStreamStep: static frame / state buffers, state updated inside the graph)
    try:
        from rvt_amd.graph import GraphedStreamStep
        model.eval()
        gs = GraphedStreamStep(model, frames[0])
        glat = []
        for i in range(8):
            gs(frames[i % 4])
        torch.cuda.synchronize()
        for i in range(max(args.steps, 50)):
            t0 = time.perf_counter()
            gs(frames[i % 4])
            torch.cuda.synchronize()
            glat.append(1e3 * (time.perf_counter() - t0))
        glat.sort()
        rec['hipgraph_wall_p50'] = round(pct(glat, 0.5), 3)
        rec['hipgraph_wall_p99'] = round(pct(glat, 0.99), 3)
        gs.close()
    except Exception as e:                                          # (reported, never fatal: the eager numbers above are the record)
        rec['hipgraph_error'] = f'{type(e).__name__}: {e}'[:200]
    if not with_detector:
        return rec
    # the whole detector step (rows f2 / f3): backbone step + YOLOX PAFPN + head + decode, inference mode, random-init weights
    from rvt_amd import fpn as _fpn, head as _head
    dims, strides = model.get_stage_dims((2, 3, 4)), model.get_strides((2, 3, 4))
    neck = _fpn.YOLOPAFPN(depth=0.67, in_channels=dims, compute_dtype=dtype).to(device).eval()
    det_head = _head.YOLOXHead(num_classes=3, strides=strides, in_channels=dims, compute_dtype=dtype).to(device).eval()
    det = []
    with torch.no_grad():
        for i in range(args.warmup + 3):
            feats, states = model(frames[i % 4], states)
            det_head(neck(feats))
        torch.cuda.synchronize()
        for i in range(max(args.steps, 50)):
            t0 = time.perf_counter()
            feats, states = model(frames[i % 4], states)
            out, _ = det_head(neck(feats))
            torch.cuda.synchronize()
            det.append(1e3 * (time.perf_counter() - t0))
    det.sort()
    rec.update({'detector_wall_p50': round(pct(det, 0.5), 3), 'detector_wall_p99': round(pct(det, 0.99), 3),
                'detector_note': 'backbone step + YOLOX PAFPN + head + decode (rvt_amd.fpn / rvt_amd.head, inference mode), same batch'})
    return rec


def also_configs(dtype_name, device, args):
    """BASELINE configs[1] (RVT-Tiny, Gen1, T=21, B=8, fwd+bwd+AdamW) and configs[4] (RVT-Base 1Mpx streaming step, B=64, T=1)
    measured in the same process right after the headline run, so that the ONE line the driver records carries them too
    (`also`).  Same step definition, same synchronisation as the headline; a few seconds in total."""
    import gc
    dtype = torch.bfloat16 if dtype_name == 'bf16' else torch.float32
    out = {}
    try:
        wl = WORKLOADS['tiny_gen1']
        model = build_model(wl, dtype, device)
        params = list(model.parameters())
        opt = torch.optim.AdamW(params, lr=2e-4, fused=True)
        xs = make_batch(wl, device, seed=1)
        T, B = wl['T'], wl['B']
        geoms = model.stage_geoms(*model.in_res_hw)
        gen = torch.Generator(device=device).manual_seed(7)
        cots = {s + 1: torch.randn((T, B, geoms[s].H, geoms[s].W, geoms[s].C), generator=gen, device=device,
                                   dtype=dtype).permute(0, 1, 4, 2, 3) for s in (1, 2, 3)}

        def step():
            feats, _ = model.forward_sequence(xs, None)
            torch.autograd.backward([feats[s] for s in (2, 3, 4)], [cots[s] for s in (2, 3, 4)])
            opt.step()
            opt.zero_grad(set_to_none=True)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        host_one = time.perf_counter() - t0
        torch.cuda.synchronize()
        gc.collect()
        K = 20
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / K
        ev = B * T / (ms * 1e-3)
        out['tiny_gen1'] = {'config': 'BASELINE configs[1]: ' + wl['label'] + f', {dtype_name}, fwd+bwd+AdamW(fused)', 'steps': K,
                            'ms_per_step': round(ms, 3), 'value': round(ev, 1), 'unit': 'event-tensors/s',
                            'mfma_frac': round(ev * wl['f_fwdbwd'] / (PEAK_TFLOPS[dtype_name] * 1e12), 4),
                            'host_enqueue_ms_per_step': round(1e3 * host_one, 2)}
        del model, params, opt, xs, cots
    except Exception as e:                         # never lose the headline to a secondary measurement
        out['tiny_gen1'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    try:
        ns = argparse.Namespace(batch=None, warmup=3, steps=100, dtype=dtype_name)
        r = stream_latency(dict(WORKLOADS['base_1mpx']), dtype, device, ns, with_detector=False)
        out['stream_latency'] = {'config': 'BASELINE configs[4]: ' + r['config']['workload'] + f', {dtype_name}, persistent ConvLSTM state',
                                 'p50': r['wall_p50'], 'p99': r['wall_p99'], 'gpu_p50': r['gpu_p50'], 'gpu_p99': r['gpu_p99'],
                                 'host_enqueue_p50': r['host_enqueue_p50'], 'unit': 'ms', 'steps': r['steps_timed'],
                                 'mfma_frac_p50': r['mfma_frac_p50'], 'event_tensors_per_s_p50': r['event_tensors_per_s_p50']}
        for k in ('hipgraph_wall_p50', 'hipgraph_wall_p99', 'hipgraph_error'):        # the same step as ONE hipGraph launch (GraphedStreamStep)
            if k in r:
                out['stream_latency'][k.replace('_wall', '')] = r[k]
    except Exception as e:
        out['stream_latency'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command under torch.distributed.run (one
    process per GPU, RCCL rendezvous on 127.0.0.1, a free port) - what the reference gets from one trainer flag
    (train.py:60-67,133).  Rank 0's JSON line goes to our stdout unchanged."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks under torch.distributed.run (127.0.0.1:{port})',
          file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='base_1mpx', choices=list(WORKLOADS))
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--batch', type=int, default=None, help='override per-GPU batch (debug only)')
    ap.add_argument('--seq', type=int, default=None, help='override T (debug only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-optimizer', action='store_true')
    ap.add_argument('--op-breakdown', default=None, help='write a per-op HIP-event time table to this path')
    ap.add_argument('--graph', default='off', choices=['auto', 'on', 'off'],
                    help='also replay the step as one captured hipGraph (rvt_amd/graph.py) and report that timing.  Off by '
                         'default: measured on MI355X / ROCm 7.2 the replay of this ~1200-node graph costs 68 ms of host time per '
                         'step (eager enqueue: 25 ms) and runs the same 106 ms on the GPU')
    ap.add_argument('--force-reducer', action='store_true', help='run the RCCL bucket all-reduce even at world size 1')
    ap.add_argument('--stream-latency', action='store_true',
                    help='BASELINE configs[4] instead of the training step: T=1 streaming inference with persistent '
                         'ConvLSTM state, batch 64; prints per-step latency percentiles (not the headline metric)')
    ap.add_argument('--no-also', action='store_true',
                    help='skip the `also` object (BASELINE configs[1] and [4] measured after the headline run)')
    ap.add_argument('--tuning', action='append', default=[], metavar='FIELD=INT',
                    help='experiment only: override a field of the RvtTuning record (the line then carries config.tuning_overrides)')
    args = ap.parse_args()
    tuning_overrides = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.tuning}
    if tuning_overrides:
        from rvt_amd import tuning as _tuning
        _tuning.use(**tuning_overrides)

    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.workload)
        return
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl['B'] = args.batch
    if args.seq:
        wl['T'] = args.seq
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))          # one process per GPU; this process only waits for them
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; pass the same N to both '
                 f'(or drop the launcher: `python bench.py --gpus N` starts the ranks itself)')
    visible = torch.cuda.device_count()
    if local_rank >= visible:
        sys.exit(f'bench.py: rank {rank} of {world} needs cuda:{local_rank} but only {visible} device(s) are visible on this box '
                 f'(one process per GPU; no CPU fallback)')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1 or args.force_reducer:
        if world == 1 and 'RANK' not in os.environ:            # --force-reducer without a launcher: a one-rank RCCL world of our own
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('nccl', device_id=device)      # "nccl" = RCCL on ROCm

    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    if args.stream_latency:
        print(json.dumps(stream_latency(wl, dtype, device, args)), flush=True)
        return
    model = build_model(wl, dtype, device)
    params = [p for p in model.parameters()]
    use_graph = args.graph == 'on' or (args.graph == 'auto' and world == 1 and not args.force_reducer)
    opt = None if args.no_optimizer else torch.optim.AdamW(params, lr=2e-4, fused=True, capturable=use_graph)
    from rvt_amd.dist import StageGradReducer
    reducer = StageGradReducer(force=args.force_reducer).attach(model) if (world > 1 or args.force_reducer) else None
    if reducer is not None:
        reducer.record_tail = True

    xs = make_batch(wl, device, seed=1 + rank)
    T, B = wl['T'], wl['B']
    geoms = model.stage_geoms(*model.in_res_hw)
    gen = torch.Generator(device=device).manual_seed(7)
    cots = {s + 1: torch.randn((T, B, geoms[s].H, geoms[s].W, geoms[s].C), generator=gen, device=device,
                               dtype=dtype).permute(0, 1, 4, 2, 3) for s in (1, 2, 3)}

    def step():
        feats, states = model.forward_sequence(xs, None)
        torch.autograd.backward([feats[s] for s in (2, 3, 4)], [cots[s] for s in (2, 3, 4)])
        if reducer is not None:
            reducer.finish()
        if opt is not None:
            opt.step()
            opt.zero_grad(set_to_none=True)
        else:
            for p in params:
                p.grad = None

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()

    # which kernel dominates?  One instrumented, untimed step with HIP events around EVERY launch; groups = (entry point, problem
    # shape), i.e. one kernel at one shape; every entry point has an algorithmic FLOP / byte model (opmodel.py), so the choice is
    # over ALL of them (round 3 could only price GEMM-family launches)
    # (the C-side training stage driver - route_stage_driver_train, the default - issues a stage's launches from inside ONE library
    # call, where this per-launch timer cannot bracket them: the instrumented passes run on the Python host loop, which launches the
    # same kernels with the same arguments in the same order, tests/test_stage_driver.py)
    from rvt_amd import tuning as _tn
    host_loop = dict(_tn.overrides(), route_stage_driver_train=0)
    timer = OpTimer()
    timer.preallocate(6000)
    timer.install()
    with _tn.override(**host_loop):
        step()
        timer.two_pass_min(step)
    prof = timer.summary()
    sustained = mfma_peak_sustained(device) if args.dtype == 'bf16' else None
    groups = kernel_groups(timer.records)
    table = [r for r in (group_roofline(k, v, args.dtype) for k, v in groups.items()) if r is not None]
    table.sort(key=lambda r: -r['ms'])
    dom_key = max((k for k in groups if opmodel.model(k[0], groups[k][0][2]) is not None),
                  key=lambda k: sum(a.elapsed_time(b) for a, b, _ in groups[k]))
    dominant = dom_key[0]
    step_ms_instrumented = sum(v['total_ms'] for v in prof.values())
    algorithmic_bytes_step = sum((opmodel.model(n, r[2]) or (0.0, 0.0))[1] for n, recs in timer.records.items() for r in recs)
    algorithmic_flops_step = sum((opmodel.model(n, r[2]) or (0.0, 0.0))[0] for n, recs in timer.records.items() for r in recs)
    executed_flops_step = sum((opmodel.executed(n, r[2]) or 0.0) for n, recs in timer.records.items() for r in recs)
    by_entry = {}
    for r in table:
        e = by_entry.setdefault(r['kernel'], [0.0, 0.0, 0.0, 0, 0.0])
        e[0] += r['ms']; e[1] += r['algorithmic_gflop']; e[2] += r['algorithmic_gbyte']; e[3] += r['launches']
        e[4] += r.get('executed_gflop', r['algorithmic_gflop'])
    entry_table = [opmodel.roofline_entry(n, e[1] * 1e9, e[2] * 1e9, e[0], e[3], args.dtype, executed_flops=e[4] * 1e9) for n, e in
                   sorted(by_entry.items(), key=lambda kv: -kv[1][0])[:10]]
    if args.op_breakdown and rank == 0:
        with open(args.op_breakdown, 'w') as f:
            tot = sum(v['total_ms'] for v in prof.values())
            f.write(f'# per-op HIP-event time of ONE step ({wl["label"]}, {args.dtype}); sum = {tot:.2f} ms\n')
            for n, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms']):
                f.write(f'{n:28s} calls={v["calls"]:5d} total={v["total_ms"]:9.3f} ms avg={v["avg_ms"]:8.4f} ms max={v["max_ms"]:8.4f} ms '
                        f'({100 * v["total_ms"] / tot:5.1f} %)\n')
            f.write('# by (op, integer args after the pointers) — which stage / shape the time goes to\n')
            by = {}
            for n, recs in timer.records.items():
                for a, b, ar in recs:
                    key = (n, tuple(x for x in ar if isinstance(x, int) and not isinstance(x, bool) and 0 <= x < (1 << 31))[-8:])
                    e = by.setdefault(key, [0, 0.0])
                    e[0] += 1
                    e[1] += a.elapsed_time(b)
            for (n, key), (cnt, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:60]:
                f.write(f'{n:28s} {str(key):60s} calls={cnt:4d} total={ms:9.3f} ms\n')
    timer.records.clear()
    timer.enabled_for = {dominant}
    top_table = table[:12]
    # two more untimed steps: the instrumented step above perturbs the caching allocator's stream-tagged pools (the
    # weight-gradient side stream), and a timed region that still grows the pool pays hipMalloc inside it
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    timer.records.clear()

    # a full (generation-2) Python GC pass over the module / autograd object graph takes 30-90 ms and lands inside one
    # step in three (seen as one 47-110 ms step among 16.6 ms ones on the Tiny workload): collect now and move the
    # survivors to the permanent generation, as any long-running training loop would
    import gc
    gc.collect()
    gc.freeze()

    def timed_region(fn):
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        marks = []
        for _ in range(args.steps):
            fn()
            m = torch.cuda.Event(enable_timing=True)
            m.record()
            marks.append(m)
        e1.record()
        host = time.perf_counter() - t0               # Python + launch time of the K steps (the GPU is still running)
        barrier()
        wall_ = time.perf_counter() - t0
        return wall_, host, [a.elapsed_time(b) for a, b in zip([e0] + marks[:-1], marks)]

    # host cost of enqueueing ONE step into an empty queue (the K-step region below measures something else on the host side:
    # with ~800 launches per step the HIP queue fills up and the host blocks on the GPU, so its host time approaches the GPU time)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host_one_step = time.perf_counter() - t0
    torch.cuda.synchronize()
    timer.records.clear()

    # eager pass: EXACTLY K steps on the production route (C-side stage drivers): the headline timing unless the graph replay below runs.
    timer.uninstall()
    if reducer is not None:
        reducer.tail_ms()                             # (drop the warm-up steps' records)
    wall, host_enqueue, per_step = timed_region(step)
    eager_ms = 1e3 * wall / args.steps
    tails = reducer.tail_ms() if reducer is not None else []
    # the `roofline` measurement: K more steps with HIP events around every launch of the dominant entry point, on the stream it is
    # launched on - on the Python host loop (same kernels, same arguments), because the stage driver hides the individual launches
    timer.install()
    with _tn.override(**host_loop):
        step()
        torch.cuda.synchronize()
        timer.records.clear()
        roof_wall, _, _ = timed_region(step)
    timer.uninstall()
    roof_pass_ms = 1e3 * roof_wall / args.steps
    if reducer is not None:
        reducer.tail_ms()
    graph_note = 'off'
    if use_graph:
        # the same K steps replayed as ONE captured hipGraph per step (no Python, no per-kernel launch cost)
        from rvt_amd.graph import GraphedStep
        try:
            gstep = GraphedStep(step, warmup=1, device=device, models=(model,))
            for _ in range(2):
                gstep()
            wall, host_enqueue, per_step = timed_region(gstep)
            graph_note = 'on'
        except Exception as e:                        # never lose the eager measurement to a capture problem
            graph_note = f'failed: {type(e).__name__}: {str(e)[:120]}'
            print(f'[bench] hipGraph capture failed, reporting the eager timing: {e}', file=sys.stderr, flush=True)
    if rank == 0:
        print(f'[bench] per-step ms: {[round(x, 1) for x in per_step]}  reserved={torch.cuda.memory_reserved() / 2**30:.1f} GiB '
              f'peak_alloc={torch.cuda.max_memory_allocated() / 2**30:.1f} GiB '
              f'alloc_retries={torch.cuda.memory_stats().get("num_alloc_retries", 0)} '
              f'host_enqueue_ms_per_step={1e3 * host_one_step:.1f} (empty queue; {1e3 * host_enqueue / args.steps:.1f} inside the timed region, '
              f'where the host blocks on a full queue) eager_ms_per_step={eager_ms:.1f} '
              f'graph={graph_note}', file=sys.stderr, flush=True)
    timer.uninstall()
    dist_info = None
    if world > 1 or args.force_reducer:
        # what RCCL itself sees, every rank's own clock, and the part of the gradient exchange the backward did not hide
        mine = torch.tensor([1e3 * wall / args.steps, (sum(tails) / len(tails)) if tails else 0.0], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        dist_info = {'world_size_rccl': dist.get_world_size(), 'backend': dist.get_backend(),
                     'per_rank_ms_per_step': [round(float(t[0]), 3) for t in allr],
                     'allreduce_tail_ms_per_rank': [round(float(t[1]), 3) for t in allr],
                     'allreduce_tail_note': 'HIP-event time from the launch of the LAST stage bucket (stage 1, the end of the backbone '
                                            'backward) to the completion of every bucket all-reduce of that step: the communication the '
                                            'backward did not hide (buckets of stages 4..2 are launched while stages 3..1 still compute)'}
    if world > 1:
        tmax = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    ms_per_step = 1e3 * wall / args.steps
    events_per_s = world * B * T / (wall / args.steps)

    if rank == 0:
        # roofline of the dominant kernel (entry point at one problem shape): algorithmic FLOPs / bytes of its launches inside the
        # timed region / the HIP-event time of the same launches
        peak = PEAK_TFLOPS[args.dtype]
        path_tflops = events_per_s / world * wl['f_fwdbwd'] / 1e12
        recs = [r for r in timer.records[dominant] if shape_key(r[2]) == dom_key[1]]
        roof = group_roofline(dom_key, recs, args.dtype)
        wkey = f'{args.workload}:{args.dtype}:B{B}:T{T}'
        dom_ms = roof['ms']
        ordered = {k: roof[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac')}
        ordered['traffic'] = traffic_lookup(wkey, dom_key)
        ordered.update({k: v for k, v in roof.items() if k not in ordered})
        ordered.update({'algorithmic_bytes_per_launch': int(roof['algorithmic_gbyte'] * 1e9 / max(len(recs), 1)),
                        'share_of_step': round(dom_ms / (roof_pass_ms * args.steps), 3),
                        'timing': 'HIP events on the launch stream around every launch of this kernel, inside a timed region of K steps that '
                                  'follows the headline region and runs the Python host loop (route_stage_driver_train = 0: same kernels, same '
                                  f'arguments; {roof_pass_ms:.3f} ms per step) - the C-side stage driver of the headline region issues a stage\'s '
                                  'launches inside one library call, where they cannot be bracketed',
                        'chosen_as': 'largest total time of one (entry point, problem shape) group over ALL entry points '
                                     '(instrumented step; opmodel.py prices every entry point)'})
        roof = ordered
        out = {
            'metric': 'event-tensors/sec (fwd+bwd) RVT-Base T=21 1Mpx; % MFMA roofline' if args.workload == 'base_1mpx'
            else 'event-tensors/sec (fwd+bwd)',
            'value': round(events_per_s, 2), 'unit': 'event-tensors/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': wl['label'], 'global_batch': world * B, 'seq_len': T,
                       'parallelism': f'dp{world}', 'optimizer': 'none' if opt is None else 'AdamW(fused)',
                       'hipgraph': graph_note, 'eager_ms_per_step': round(eager_ms, 3),
                       'host_enqueue_ms_per_step': round(1e3 * host_one_step, 2),
                       'host_ms_per_step_in_timed_region': round(1e3 * host_enqueue / args.steps, 2),
                       'upstream_grads': 'random cotangents on stage 2-4 features of all T frames',
                       **({'tuning_overrides': tuning_overrides} if tuning_overrides else {})},
            'mfma_roofline_frac_whole_step': round(path_tflops / peak, 4),
            'algorithmic_tflops_per_gpu': round(path_tflops, 2),
            'roofline': roof,
            # the 12 heaviest kernels (entry point at one shape) and the 10 heaviest entry points of ONE instrumented step
            # (HIP events around every launch, one stream, untimed): ms, algorithmic GFLOP / GB, which roof binds, fraction of it
            'roofline_table': top_table,
            'roofline_entry_points': entry_table,
            'instrumented_step_ms': round(step_ms_instrumented, 3),
            'instrumented_step_note': 'sum over launches of min(duration in pass 1, duration in pass 2) of two instrumented steps: a lower '
                                      'envelope, not comparable with eager_ms_per_step (a mean over K un-instrumented steps)',
            'hbm_traffic_per_step': {'measured_bytes': step_traffic(wkey), 'source': 'profiles/latest_traffic.json (rocprofv3 FETCH_SIZE x2 + '
                                     'WRITE_SIZE passes of this command), null when no pass of THIS workload is committed',
                                     # what the launches of one step move if every operand / result crosses HBM exactly once (opmodel.py, sum
                                     # over the instrumented step) and what a fully fused stage-by-stage path would (19.4 MB per
                                     # event tensor forward, x3 for forward + backward)
                                     'algorithmic_gbyte_per_step': round(algorithmic_bytes_step / 1e9, 2),
                                     'fused_minimum_gbyte_per_step': round(19.4e6 * B * T * 3 / 1e9, 2) if args.workload == 'base_1mpx' else None},
            'flops_per_step': {'algorithmic_gflop_of_the_launches': round(algorithmic_flops_step / 1e9, 1),
                               'executed_gflop_incl_recompute': round(executed_flops_step / 1e9, 1),
                               'survey_8d_gflop': round(wl['f_fwdbwd'] * B * T / 1e9, 1)},
            'mfma_peak_sustained': None if not sustained else {
                'tflops': round(sustained, 1), 'implied_clock_ghz': round(sustained * 1e12 / (256 * 4096) / 1e9, 3),
                'whole_step_frac_of_sustained': round(path_tflops / sustained, 4),
                'method': 'rvt_probe_mfma: 4096 workgroups of nothing but independent v_mfma_f32_32x32x16_bf16, HIP-event timed'},
        }
        if dist_info is not None:
            out['distributed'] = dist_info
        if world == 1 and args.workload == 'base_1mpx' and not args.no_also and not (args.batch or args.seq):
            del xs, cots
            model = opt = params = None
            gc.collect()
            torch.cuda.empty_cache()
            out['also'] = also_configs(args.dtype, device, args)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_reducer:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
