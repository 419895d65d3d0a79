#!/bin/bash
# detection-tail check: the FPN / head / SimOTA GPU tests and the detection-tail micro-benchmark with its kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4d
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_head.py tests/test_fpn.py -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest_head.log; cat $OUT/pytest_head.log
timeout 600 python profiles/microbench_detect.py 48 16 > $OUT/microbench_detect.txt 2> $OUT/microbench_detect.err; tail -3 $OUT/microbench_detect.err; cat $OUT/microbench_detect.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_detect/trace -o detect -- python $ROOT/profiles/microbench_detect.py 48 16 > /dev/null 2> $OUT/rocprof_detect.err
cd $ROOT
python profiles/summarize_rocprof.py $OUT/prof_detect > $OUT/detect_kernel_summary.txt 2>&1; head -45 $OUT/detect_kernel_summary.txt
rm -rf $OUT/prof_detect
