#!/bin/bash
# round-3 evidence: GPU parity tests, the three bench lines (default = BASELINE configs[2], tiny_gen1 = configs[1], streaming = configs[4]),
# per-op table, rocprofv3 kernel stats + FETCH / WRITE passes of the bench command, micro-benchmarks of the round's kernels
# (ppgemm / ppgemm_tn against the 128-row engine, conv dgrad4, chain MLP), PMC counters of the ppgemm kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3i}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 300 python bench.py --workload tiny_gen1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_tiny_gen1.json 2> $OUT/bench_tiny_gen1.err
tail -1 $OUT/bench_tiny_gen1.err; cut -c1-200 $OUT/bench_tiny_gen1.json
timeout 300 python bench.py --stream-latency --steps 100 --warmup 5 > $OUT/bench_stream_latency.json 2> $OUT/bench_stream_latency.err
tail -1 $OUT/bench_stream_latency.err; cut -c1-300 $OUT/bench_stream_latency.json
timeout 600 python bench.py --steps 20 --warmup 5 --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.err; cat $OUT/bench_default.json
bash profiles/run_rocprof.sh $TAG 2>&1 | tail -3
RVT_PPGEMM=1 timeout 600 python profiles/microbench_ppgemm.py > $OUT/microbench_ppgemm.txt 2>&1
RVT_PPGEMM=0 timeout 600 python profiles/microbench_ppgemm.py > $OUT/microbench_gemm128.txt 2>&1
timeout 300 python profiles/microbench_conv_dgrad.py > $OUT/microbench_conv_dgrad.txt 2>&1
timeout 300 python profiles/microbench_mlp_chain.py > $OUT/microbench_mlp_chain.txt 2>&1
timeout 300 python profiles/microbench_dgrad_ln.py > $OUT/microbench_dgrad_ln.txt 2>&1
tail -n 2 $OUT/microbench_ppgemm.txt $OUT/microbench_conv_dgrad.txt $OUT/microbench_mlp_chain.txt $OUT/microbench_dgrad_ln.txt
bash profiles/pmc_probe.sh pp_fwd_s4 pp_dgrad_s4 pp_scale_res_s4 pp_wgrad_s4 pp_fwd_s3 pp_wgrad_s3 conv_dgrad4_s3 mlpc_fwd mlpc_dgrad mlpc_wgrad dgrad_ln_k512 dgrad_ln_k384 > $OUT/pmc_round3_kernels.txt 2>&1
grep -c "==" $OUT/pmc_round3_kernels.txt
