#!/bin/bash
# round 6, call e: lstm_scan3 at C = 128 too (stage 2 of RVT-Base, stage 3 of RVT-Tiny): tests + same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests/test_lstm_scan.py tests/test_production_route.py tests/test_backbone.py tests/test_opmodel.py tests/test_step.py -x -q -m gpu > gpurun_out/r6e/pytest.log 2>&1; tail -4 gpurun_out/r6e/pytest.log
for v in 3 1 3 1; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --tuning lstm_scan3=$v --op-breakdown gpurun_out/r6e/op_$v.txt > gpurun_out/r6e/bench_$v.json 2> gpurun_out/r6e/bench_$v.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6e/bench_$v.json') if l.startswith('{')][-1]); print('lstm_scan3=$v', d['ms_per_step'], d['value'], d['also']['tiny_gen1']['ms_per_step'])"
done
