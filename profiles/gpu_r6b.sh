#!/bin/bash
# round 6, call b: same-box A/B of -ffp-contract=off (shipped) vs fast
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6b
for rep in 1 2; do
for v in base fc; do
  if [ $v = fc ]; then export RVT_HIP_LIB=$PWD/rvt_amd/librvt_hip_fc.so; else unset RVT_HIP_LIB; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --op-breakdown gpurun_out/r6b/op_$v.txt > gpurun_out/r6b/bench_${v}_$rep.json 2> gpurun_out/r6b/bench_${v}_$rep.err
  python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r6b/bench_${v}_$rep.json') if l.startswith('{')][-1]); print('$v', $rep, d['ms_per_step'], d['value'])"
done; done
