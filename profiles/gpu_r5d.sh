#!/bin/bash
# round 5: ln_linear unit tests + micro-benchmark + same-box A/B of the step with / without it.  usage: gpu_r5d.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5d}
mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "ln_linear" > $OUT/pytest_ln_linear.log 2>&1; tail -3 $OUT/pytest_ln_linear.log
timeout 300 python profiles/microbench_ln_linear.py > $OUT/microbench_ln_linear.txt 2>&1; cat $OUT/microbench_ln_linear.txt
timeout 1500 python -m pytest tests/test_backbone.py tests/test_production_route.py tests/test_host.py -m gpu -x -q > $OUT/pytest_backbone.log 2>&1; tail -3 $OUT/pytest_backbone.log
bash profiles/gpu_r5c.sh ${1:-r5d} "ln_linear=0"
