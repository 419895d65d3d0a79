#!/bin/bash
# kernel-trace stats of the default bench step (top kernels only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-trace_base}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
python $ROOT/profiles/summarize_rocprof.py $OUT 2>/dev/null | head -${2:-45}
find $OUT -name '*.db' -delete; find $OUT -name '*.csv' -size +4M -delete
