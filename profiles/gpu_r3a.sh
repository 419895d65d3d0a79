#!/bin/bash
# round 3, first GPU pass: the 256 x 256 LDS-DMA GEMM (csrc/ppgemm.hpp) on the hardware - parity tests through the C ABI,
# A/B micro-benchmark against the 128-row engine on the stage-3/4 shapes, PMC counters of two of its launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3a}
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -x -k "ppgemm or linear" 2>&1 | tail -2; done > $OUT/pytest_ppgemm.log; cat $OUT/pytest_ppgemm.log
RVT_PPGEMM=1 timeout 600 python profiles/microbench_ppgemm.py > $OUT/microbench_ppgemm_new.txt 2>&1; cat $OUT/microbench_ppgemm_new.txt
RVT_PPGEMM=0 timeout 600 python profiles/microbench_ppgemm.py > $OUT/microbench_ppgemm_old.txt 2>&1; cat $OUT/microbench_ppgemm_old.txt
bash profiles/pmc_probe.sh fwd_k512 dgrad_k1024 2>&1 | tail -40 > $OUT/pmc_ppgemm.txt; cat $OUT/pmc_ppgemm.txt
