#!/bin/bash
# copy the outputs of profiles/gpu_r6_evidence.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked).  usage: profiles/collect_r6.sh <tag>
set -e
TAG=${1:-r6a}
cd "$(dirname "$0")/.."
G=gpurun_out/$TAG; P=gpurun_out/prof_$TAG
mkdir -p profiles/${TAG}_rocprofv3 profiles/r6
cp $P/summary_$TAG.txt profiles/${TAG}_rocprofv3/summary.txt
cp $P/traffic.json profiles/${TAG}_rocprofv3/traffic.json
cp $P/traffic.json profiles/latest_traffic.json
cp $P/trace/trace_kernel_stats.csv profiles/${TAG}_rocprofv3/kernel_stats.csv
cp $G/bench_default.json profiles/bench_${TAG}_base_1mpx_n1.json.log
cp $G/bench_tiny_gen1.json profiles/bench_${TAG}_tiny_gen1_n1.json.log
cp $G/bench_stream_latency.json profiles/bench_${TAG}_stream_latency.json.log
cp $G/bench_reducer_world1.json profiles/bench_${TAG}_reducer_world1.json.log
cp $G/op_breakdown.txt profiles/r6/op_breakdown_$TAG.txt
cp $G/op_breakdown_tiny.txt profiles/r6/op_breakdown_tiny_$TAG.txt
cp $G/pytest.log profiles/r6/pytest_gpu_$TAG.log
ls -la profiles/${TAG}_rocprofv3 profiles/r6 | tail -30
