#!/bin/bash
# fused attention half: parity tests on the GPU + micro-benchmark against the op-by-op chain
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2e
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "attn_block or attention" 2>&1 | tail -8 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python profiles/microbench_attn_block.py > $OUT/microbench_attn_block.txt 2>&1
cat $OUT/microbench_attn_block.txt
