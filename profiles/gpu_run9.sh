#!/bin/bash
# GPU parity tests + bench line + per-op table (single stream = isolated kernel times)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2l}
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_base.json 2> $OUT/bench_base.err
tail -1 $OUT/bench_base.err; cat $OUT/bench_base.json | cut -c1-330; head -30 $OUT/op_breakdown.txt
