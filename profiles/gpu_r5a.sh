#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5a
mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_full.log 2>&1; tail -4 $OUT/pytest_full.log > $OUT/pytest.log; grep -B30 "short test summary" $OUT/pytest_full.log | head -60 > $OUT/pytest_fail.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -2 $OUT/pytest.log; cut -c1-400 $OUT/bench_default.json; head -30 $OUT/op_breakdown.txt
