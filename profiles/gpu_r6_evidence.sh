#!/bin/bash
# round-6 evidence script: GPU parity tests, the bench line (with `also`: BASELINE configs[1] / [4]) + per-op table, the two secondary
# bench lines on their own, rocprofv3 kernel stats + FETCH / WRITE passes of the bench command.  Usage: bash profiles/gpu_r6_evidence.sh <tag> [quick]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r6a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -2 $OUT/bench_default.err; cut -c1-1200 $OUT/bench_default.json
if [ "$2" != "quick" ]; then
timeout 300 python bench.py --workload tiny_gen1 --steps 20 --warmup 5 --no-cpu-baseline --op-breakdown $OUT/op_breakdown_tiny.txt > $OUT/bench_tiny_gen1.json 2> $OUT/bench_tiny_gen1.err
tail -1 $OUT/bench_tiny_gen1.err; cut -c1-200 $OUT/bench_tiny_gen1.json
timeout 300 python bench.py --stream-latency --steps 100 --warmup 5 > $OUT/bench_stream_latency.json 2> $OUT/bench_stream_latency.err
tail -1 $OUT/bench_stream_latency.err; cut -c1-300 $OUT/bench_stream_latency.json
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-also --force-reducer > $OUT/bench_reducer_world1.json 2> $OUT/bench_reducer_world1.err
cut -c1-200 $OUT/bench_reducer_world1.json
bash profiles/run_rocprof.sh $TAG 2>&1 | tail -3
fi
head -40 $OUT/op_breakdown.txt
