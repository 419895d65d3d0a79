#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3e}
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -x -k "ppgemm or linear or lstm_cell" 2>&1 | tail -2; done > $OUT/pytest_ppgemm.log; cat $OUT/pytest_ppgemm.log
RVT_PPGEMM=1 timeout 600 python profiles/microbench_ppgemm.py 2>&1 | grep "wgrad\|!!" > $OUT/microbench_wgrad_new.txt; cat $OUT/microbench_wgrad_new.txt
RVT_PPGEMM=0 timeout 600 python profiles/microbench_ppgemm.py 2>&1 | grep "wgrad\|!!" > $OUT/microbench_wgrad_old.txt; cat $OUT/microbench_wgrad_old.txt
