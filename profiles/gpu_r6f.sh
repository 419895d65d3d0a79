#!/bin/bash
# round 6, call f: full GPU suite + bench with lstm_scan3 at C = 128 / 256 (defaults)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6f
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6f/pytest.log 2>&1; tail -4 gpurun_out/r6f/pytest.log
python bench.py --steps 20 --warmup 3 --op-breakdown gpurun_out/r6f/op.txt > gpurun_out/r6f/bench.json 2> gpurun_out/r6f/bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6f/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['mfma_roofline_frac_whole_step'], json.dumps(d['also'])[:600]); print(json.dumps(d['roofline'])[:400])"
