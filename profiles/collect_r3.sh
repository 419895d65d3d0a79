#!/bin/bash
# copy the outputs of profiles/gpu_r3_final.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked).  usage: profiles/collect_r3.sh <tag>
set -e
TAG=${1:-r3i}
cd "$(dirname "$0")/.."
G=gpurun_out/$TAG; P=gpurun_out/prof_$TAG
python profiles/summarize_rocprof.py $P > $P/summary_$TAG.txt
mkdir -p profiles/${TAG}_rocprofv3 profiles/r3
cp $P/summary_$TAG.txt profiles/${TAG}_rocprofv3/summary.txt
cp $P/traffic.json profiles/${TAG}_rocprofv3/traffic.json
cp $P/traffic.json profiles/latest_traffic.json
cp $P/trace/trace_kernel_stats.csv profiles/${TAG}_rocprofv3/kernel_stats.csv
cp $G/bench_default.json profiles/bench_${TAG}_base_1mpx_n1.json.log
cp $G/bench_tiny_gen1.json profiles/bench_${TAG}_tiny_gen1_n1.json.log
cp $G/bench_stream_latency.json profiles/bench_${TAG}_stream_latency.json.log
cp $G/op_breakdown.txt profiles/r3/op_breakdown_$TAG.txt
cp $G/pytest.log profiles/r3/pytest_gpu_$TAG.log
for f in microbench_ppgemm microbench_gemm128 microbench_conv_dgrad microbench_mlp_chain microbench_dgrad_ln pmc_round3_kernels; do cp $G/$f.txt profiles/r3/${f}_$TAG.txt; done
python profiles/pmc_summary.py pp_fwd_s4 pp_dgrad_s4 pp_scale_res_s4 pp_wgrad_s4 pp_fwd_s3 pp_wgrad_s3 conv_dgrad4_s3 mlpc_fwd mlpc_dgrad mlpc_wgrad dgrad_ln_k512 dgrad_ln_k384 > profiles/r3/pmc_round3_table_$TAG.txt
ls -la profiles/${TAG}_rocprofv3 profiles/r3 | tail -30
