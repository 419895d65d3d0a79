#!/bin/bash
# round 5: conv weight gradient on ppgemm_tn (CONV): unit tests + micro-benchmark + detector / fpn / backbone tests + same-box A/B.  usage: gpu_r5e.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5e}
mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv" > $OUT/pytest_conv.log 2>&1; tail -3 $OUT/pytest_conv.log
timeout 300 python profiles/microbench_conv_wgrad.py > $OUT/microbench_conv_wgrad.txt 2>&1; cat $OUT/microbench_conv_wgrad.txt
timeout 1500 python -m pytest tests/test_backbone.py tests/test_production_route.py tests/test_fpn.py tests/test_detector_step.py -m gpu -x -q > $OUT/pytest_backbone.log 2>&1; tail -3 $OUT/pytest_backbone.log
bash profiles/gpu_r5c.sh ${1:-r5e} "conv_wgrad_tn=0" "conv_wgrad_tn=0 ln_linear=0"
