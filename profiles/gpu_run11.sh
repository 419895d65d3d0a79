#!/bin/bash
# round-2 final evidence: GPU parity tests, the three bench lines (default = BASELINE configs[2], tiny_gen1 = configs[1], streaming = configs[4]),
# per-op table, rocprofv3 kernel stats + FETCH / WRITE passes, PMC counters of the kernels added in the second half of the round
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2v}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 300 python bench.py --workload tiny_gen1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_tiny_gen1.json 2> $OUT/bench_tiny_gen1.err
tail -1 $OUT/bench_tiny_gen1.err; cut -c1-200 $OUT/bench_tiny_gen1.json
timeout 300 python bench.py --stream-latency --steps 100 --warmup 5 > $OUT/bench_stream_latency.json 2> $OUT/bench_stream_latency.err
tail -1 $OUT/bench_stream_latency.err; cut -c1-300 $OUT/bench_stream_latency.json
timeout 600 python bench.py --steps 20 --warmup 5 --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.err; cat $OUT/bench_default.json
bash profiles/run_rocprof.sh ${1:-r2v} 2>&1 | tail -3
bash profiles/pmc_probe.sh mlpc_wgrad stem_fwd stem_wgrad 2>&1 | tail -40
