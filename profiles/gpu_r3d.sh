#!/bin/bash
# full GPU parity suite + default bench line with the per-op table
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3d}
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -2 $OUT/bench_default.err; cat $OUT/bench_default.json; head -45 $OUT/op_breakdown.txt
