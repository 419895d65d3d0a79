#!/bin/bash
# round-2 state check: GPU parity tests, bench line with op table, rocprofv3 trace + PMC passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2d
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 400 python bench.py --steps 10 --warmup 3 --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_base.json 2> $OUT/bench_base.err
tail -2 $OUT/bench_base.err; cat $OUT/bench_base.json; head -40 $OUT/op_breakdown.txt
bash profiles/run_rocprof.sh r2d > $OUT/rocprof.log 2>&1
cat $ROOT/gpurun_out/prof_r2d/summary_r2d.txt | head -70
