#!/bin/bash
# round 5: TN weight-gradient item mapping (slices straddling XCDs): kernel tests + conv micro-benchmark + op breakdown of the step.  usage: gpu_r5f.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5f}
mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q > $OUT/pytest_kernels.log 2>&1; tail -3 $OUT/pytest_kernels.log
timeout 300 python profiles/microbench_conv_wgrad.py > $OUT/microbench_conv_wgrad.txt 2>&1; cat $OUT/microbench_conv_wgrad.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --op-breakdown $OUT/op_breakdown.txt > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import json;d=json.load(open('$OUT/bench_default.json'));print('ms_per_step', d['ms_per_step'])"
grep -E "rvt_linear_wgrad|rvt_lstm_wgrad|rvt_conv_wgrad|rvt_ln_linear" $OUT/op_breakdown.txt
