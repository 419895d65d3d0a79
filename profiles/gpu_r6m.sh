#!/bin/bash
# round 6, call m: C-side training stage driver: parity + same-box A/B (host enqueue, step time)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6m
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6m/pytest.log 2>&1; tail -4 gpurun_out/r6m/pytest.log
for v in 1 0 1 0; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --tuning route_stage_driver_train=$v > gpurun_out/r6m/bench_$v.json 2> gpurun_out/r6m/bench_$v.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6m/bench_$v.json') if l.startswith('{')][-1]); a=d['also']; print('train driver=$v', d['ms_per_step'], d['value'], 'host', d['config']['host_enqueue_ms_per_step'], '| tiny', a['tiny_gen1']['ms_per_step'], 'host', a['tiny_gen1']['host_enqueue_ms_per_step'])"
done
