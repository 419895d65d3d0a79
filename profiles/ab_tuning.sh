#!/bin/bash
# Same-box A/B of ONE tuning field (0 / 1) with the tree's library: GPU parity tests, then bench.py --tuning FIELD=0|1 twice each and the per-op differences.
#   usage (through gpurun): bash profiles/ab_tuning.sh <field>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abt
FIELD=${1:-conv_fwd_pp}
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --tuning $FIELD=$v --op-breakdown gpurun_out/abt/op_$v.txt > gpurun_out/abt/bench_${v}_$rep.json 2> gpurun_out/abt/bench_${v}_$rep.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/abt/bench_${v}_$rep.json') if l.startswith('{')][-1]); print('$FIELD=$v', $rep, d['ms_per_step'], d['value'])"
done; done 2>&1 | tee gpurun_out/abt/ab.txt
python - <<PY
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r'(\S+)\s+calls=\s*(\d+)\s+total=\s*([\d.]+)',l)
        if m: d[m.group(1)]=float(m.group(3))
    return d
a,b=load('gpurun_out/abt/op_0.txt'),load('gpurun_out/abt/op_1.txt')
for k in sorted(a,key=lambda k:-abs(a[k]-b.get(k,0)))[:6]: print('%-32s off %7.3f on %7.3f  %+.3f'%(k,a[k],b.get(k,0),b.get(k,0)-a[k]))
PY
grep -h "rvt_conv_fwd  " gpurun_out/abt/op_0.txt gpurun_out/abt/op_1.txt
