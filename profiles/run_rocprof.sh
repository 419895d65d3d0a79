#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun).  Pass 1: kernel trace + stats.
# Passes 2/3: HBM byte counters in their own runs (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass;
# never combined with sys/hip/hsa tracing).  Usage: profiles/run_rocprof.sh <tag>
set -x
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
find $OUT -type f | head -50
# keep only what fits the 64 MiB pull limit: stats + compacted per-kernel aggregates
python $ROOT/profiles/summarize_rocprof.py $OUT base_1mpx:bf16:B24:T21 11 > $OUT/summary_$TAG.txt 2>&1      # (11 steps = warmup 1 + instrumented 2 + untimed 2 + host-enqueue probe 1 + timed 2 + roofline pass 1 + 2)
cat $OUT/summary_$TAG.txt | head -80
find $OUT -name '*.db' -delete
find $OUT -name '*.csv' -size +4M -delete
tail -5 $OUT/trace.log
du -sh $OUT
