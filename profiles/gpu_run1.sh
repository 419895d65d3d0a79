#!/bin/bash
# GPU session r2a: parity tests, bench lines (eager + hipGraph), Tiny / streaming lines, drift report, kernel trace.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2a
mkdir -p $OUT
cd $ROOT
export RVT_DRIFT_REPORT=$OUT/drift.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > $OUT/bench_base.json 2> $OUT/bench_base.err
tail -3 $OUT/bench_base.err; cat $OUT/bench_base.json
timeout 200 python bench.py --steps 10 --warmup 3 --workload tiny_gen1 --no-cpu-baseline > $OUT/bench_tiny.json 2> $OUT/bench_tiny.err
tail -2 $OUT/bench_tiny.err; cat $OUT/bench_tiny.json
timeout 200 python bench.py --stream-latency --steps 100 --warmup 5 > $OUT/bench_stream.json 2> $OUT/bench_stream.err
cat $OUT/bench_stream.json
timeout 200 python bench.py --steps 6 --warmup 2 --graph off --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_eager.json 2> $OUT/bench_eager.err
tail -2 $OUT/bench_eager.err; head -25 $OUT/op_breakdown.txt
