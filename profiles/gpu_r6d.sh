#!/bin/bash
# round 6, call d: wide-stage ConvLSTM scan (lstm_scan3) in the step: tests on the production route + same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6d
timeout 900 python -m pytest tests/test_lstm_scan.py tests/test_production_route.py tests/test_backbone.py tests/test_opmodel.py tests/test_step.py -x -q -m gpu > gpurun_out/r6d/pytest.log 2>&1; tail -4 gpurun_out/r6d/pytest.log
timeout 300 python profiles/microbench_lstm_scan3.py 2>&1 | tee gpurun_out/r6d/microbench_scan3.txt
for v in on off; do
  if [ $v = off ]; then T="--tuning lstm_scan3=0"; else T=""; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline $T --op-breakdown gpurun_out/r6d/op_$v.txt > gpurun_out/r6d/bench_$v.json 2> gpurun_out/r6d/bench_$v.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6d/bench_$v.json') if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['value'], json.dumps(d.get('also'))[:400])"
done
