#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2g
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "mlp" 2>&1 | tail -5 > $OUT/pytest.log
cat $OUT/pytest.log
RVT_MLP_CHAIN=0 timeout 200 python profiles/microbench_mlp_chain.py > $OUT/microbench_mlp_chain.txt 2>&1
timeout 200 python profiles/microbench_mlp_chain.py >> $OUT/microbench_mlp_chain.txt 2>&1
cat $OUT/microbench_mlp_chain.txt
