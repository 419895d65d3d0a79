#!/bin/bash
# round 5: streamed C = 128 MLP kernels — parity on the GPU + micro-benchmark
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5b}
mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "mlp" 2>&1 | tail -5
timeout 300 python profiles/microbench_mlp_stream.py 2>&1 | tee $OUT/microbench_mlp_stream.txt
