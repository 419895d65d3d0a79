#!/bin/bash
# quick check after a kernel change: GPU tests, bench line with per-op table
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4f}
mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.err | cut -c1-330; cut -c1-330 $OUT/bench_default.json; head -14 $OUT/op_breakdown.txt
