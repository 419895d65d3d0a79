#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_stem_ea; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT -o pmc -- python $ROOT/profiles/pmc_probe.py stem_fwd $PROBE_LIB > $OUT/log.txt 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o pmc -- python $ROOT/profiles/pmc_probe.py stem_fwd $PROBE_LIB > $OUT/logf.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(float)); n=defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1],'**','*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:50]
        if 'stem' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
for k,c in agg.items():
    print(k)
    for name,v in c.items(): print('   ',name, v/n[k][name])
    if 'TCC_EA0_RDREQ_sum' in c:
        tot=c['TCC_EA0_RDREQ_sum']/n[k]['TCC_EA0_RDREQ_sum']; b32=c['TCC_EA0_RDREQ_32B_sum']/n[k]['TCC_EA0_RDREQ_32B_sum']
        print('    => fabric read bytes per launch (32B x n32 + 64B x rest): %.3f GB' % ((b32*32+(tot-b32)*64)/1e9))
    if 'FETCH_SIZE' in c: print('    => FETCH_SIZE KiB*1024: %.3f GB raw, x2 = %.3f GB' % (c['FETCH_SIZE']/n[k]['FETCH_SIZE']*1024/1e9, c['FETCH_SIZE']/n[k]['FETCH_SIZE']*2048/1e9))
PY
tail -2 $OUT/log.txt
