#!/bin/bash
# round 6, call a: bench self-launch + `also` on the GPU, side-stream lifetime test, same-box baseline of the r5 kernels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6a
python -m pytest tests/test_dist_gpu.py tests/test_production_route.py -x -q -m gpu -k "bench or side_stream or deferred" > gpurun_out/r6a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6a/pytest.log
python bench.py --steps 20 --warmup 3 --op-breakdown gpurun_out/r6a/op_breakdown.txt > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
echo "bench rc=$?" >> gpurun_out/r6a/bench.err
tail -3 gpurun_out/r6a/pytest.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6a/bench.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['ms_per_step'], d['value'], d['mfma_roofline_frac_whole_step'], json.dumps(d.get('also')), json.dumps(d.get('cpu_baseline'))[:300])
PY
