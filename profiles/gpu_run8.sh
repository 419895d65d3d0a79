#!/bin/bash
# serial accounting: the whole step on ONE stream (no weight-gradient side stream) with per-op HIP-event times = isolated times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2i}
mkdir -p $OUT
cd $ROOT
RVT_WGRAD_STREAM=0 timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --op-breakdown $OUT/op_breakdown_serial.txt $BENCH_ARGS > $OUT/bench_serial.json 2> $OUT/bench_serial.err
tail -1 $OUT/bench_serial.err; head -32 $OUT/op_breakdown_serial.txt
