#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2c
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_base.json 2> $OUT/bench_base.err
tail -2 $OUT/bench_base.err; cat $OUT/bench_base.json; head -28 $OUT/op_breakdown.txt; grep -E "mlp|lstm_scan" $OUT/op_breakdown.txt
