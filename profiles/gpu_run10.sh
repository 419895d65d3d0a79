#!/bin/bash
# BASELINE configs[1] (RVT-Tiny, Gen1, T=21, B=8) and configs[4] (streaming inference, T=1, B=64) bench lines; default bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2n}
mkdir -p $OUT
cd $ROOT
timeout 300 python bench.py --workload tiny_gen1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_tiny_gen1.json 2> $OUT/bench_tiny_gen1.err
tail -1 $OUT/bench_tiny_gen1.err; cat $OUT/bench_tiny_gen1.json
timeout 300 python bench.py --stream-latency --steps 100 --warmup 5 > $OUT/bench_stream_latency.json 2> $OUT/bench_stream_latency.err
tail -1 $OUT/bench_stream_latency.err; cat $OUT/bench_stream_latency.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.err; cat $OUT/bench_default.json
