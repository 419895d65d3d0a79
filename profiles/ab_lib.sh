#!/bin/bash
# Same-box A/B of two builds of the library: the tree's rvt_amd/librvt_hip.so ("new") against rvt_amd/librvt_hip_<suffix>.so ("base",
# profiles/build_variant.sh).  usage: bash profiles/ab_lib.sh <tag> [suffix] [microbench.py ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
TAG=${1:-ab}; SUF=${2:-base}; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for mb in "$@"; do
  for v in base new; do
    if [ $v = base ]; then export RVT_HIP_LIB=$ROOT/rvt_amd/librvt_hip_$SUF.so; else unset RVT_HIP_LIB; fi
    echo "== $mb [$v]"; timeout 300 python $mb 2>&1 | tail -6
  done
done 2>&1 | tee $OUT/microbench.txt
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export RVT_HIP_LIB=$ROOT/rvt_amd/librvt_hip_$SUF.so; else unset RVT_HIP_LIB; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --op-breakdown $OUT/op_$v.txt > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/bench_${v}_$rep.json') if l.startswith('{')][-1]); print('$v', $rep, d['ms_per_step'], d['value'])"
done; done 2>&1 | tee $OUT/ab.txt
python - <<PY
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r'(\S+)\s+calls=\s*(\d+)\s+total=\s*([\d.]+)',l)
        if m: d[m.group(1)]=float(m.group(3))
    return d
a,b=load('$OUT/op_base.txt'),load('$OUT/op_new.txt')
for k in sorted(a,key=lambda k:-abs(a[k]-b.get(k,0)))[:8]: print('%-32s base %7.3f new %7.3f  %+.3f'%(k,a[k],b.get(k,0),b.get(k,0)-a[k]))
PY
