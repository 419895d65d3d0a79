#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b
mkdir -p $OUT
cd $ROOT
export RVT_DRIFT_REPORT=$OUT/drift.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_base.json 2> $OUT/bench_base.err
tail -2 $OUT/bench_base.err; cat $OUT/bench_base.json; head -24 $OUT/op_breakdown.txt; grep lstm $OUT/op_breakdown.txt
RVT_LSTM_SCAN=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --op-breakdown $OUT/op_breakdown_scan128.txt > $OUT/bench_scan128.json 2> $OUT/bench_scan128.err
tail -2 $OUT/bench_scan128.err; grep lstm $OUT/op_breakdown_scan128.txt
cat $OUT/drift.txt
