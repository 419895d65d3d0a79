#!/bin/bash
# kernel trace of the RVT-Tiny / Gen1 step (what bounds the small configuration)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3h}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload tiny_gen1 --steps 3 --warmup 2 --no-cpu-baseline"
$BENCH > $OUT/bench_tiny.log 2>&1; tail -1 $OUT/bench_tiny.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
find $OUT -name '*kernel_stats.csv' | head -1 | xargs -I{} head -45 {} > $OUT/tiny_kernel_stats_top.csv
cat $OUT/tiny_kernel_stats_top.csv | cut -c1-200
find $OUT -name '*.db' -delete
find $OUT -name '*.csv' -size +4M -delete
