#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6h
for rep in 1 2; do
for v in base var; do
  if [ $v = var ]; then export RVT_HIP_LIB=$PWD/rvt_amd/librvt_hip_w.so; else unset RVT_HIP_LIB; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --op-breakdown gpurun_out/r6h/op_$v.txt > gpurun_out/r6h/bench_${v}_$rep.json 2> gpurun_out/r6h/bench_${v}_$rep.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r6h/bench_${v}_$rep.json') if l.startswith('{')][-1]); print('$v', $rep, d['ms_per_step'], d['value'])"
done; done
