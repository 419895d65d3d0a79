#!/bin/bash
# round-4 final evidence: everything profiles/gpu_r4a.sh collects (GPU tests, three bench lines, per-op table, rocprofv3 trace +
# FETCH / WRITE passes), the detection-tail micro-benchmark with its kernel trace, and the PMC pass of the rebuilt ConvLSTM scan.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4e}
cd $ROOT
bash profiles/gpu_r4a.sh $TAG
OUT=$ROOT/gpurun_out/$TAG
timeout 600 python profiles/microbench_detect.py 48 16 > $OUT/microbench_detect.txt 2> $OUT/microbench_detect.err; tail -2 $OUT/microbench_detect.err; tail -1 $OUT/microbench_detect.txt
timeout 900 bash profiles/pmc_probe.sh scan_fwd_s1 scan_bwd_s1 > $OUT/pmc_scan2.txt 2>&1; tail -30 $OUT/pmc_scan2.txt
