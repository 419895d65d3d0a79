#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3b}
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -x -k "ppgemm or linear" 2>&1 | tail -2; done > $OUT/pytest_ppgemm.log; cat $OUT/pytest_ppgemm.log
(cd profiles/probes && ./ppgemm_probe 120960 2048 512 && ./ppgemm_probe 483840 256 1024 && ./ppgemm_probe 483840 1024 256) > $OUT/probe.txt 2>&1; cat $OUT/probe.txt
RVT_PPGEMM=1 timeout 600 python profiles/microbench_ppgemm.py > $OUT/microbench_ppgemm_new.txt 2>&1; cat $OUT/microbench_ppgemm_new.txt
