#!/bin/bash
# SQ issue/stall counters of one isolated op (own pass, kernel-trace only).  usage: profiles/pmc_probe.sh <op> [<op> ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for OP in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_$OP; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $OUT -o pmc -- python $ROOT/profiles/pmc_probe.py $OP > $OUT/log.txt 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/L2 -o pmc -- python $ROOT/profiles/pmc_probe.py $OP > $OUT/log_L2.txt 2>&1
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/$CTR -o pmc -- python $ROOT/profiles/pmc_probe.py $OP > $OUT/log_$CTR.txt 2>&1
  done
  python - "$OUT" "$OP" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, op = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rvt' not in k: continue
        k = k[:60] + ('|' + k.split('Ep')[-1][:16] if 'Ep' in k else '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'): agg[k][r['Counter_Name'] + '_dispatches'] += 1
for k, c in agg.items():
    d = max(n[k], 1)
    print(f'== {op}: {k}  dispatches={n[k]}')
    wc = c.get('SQ_WAVE_CYCLES', 1.0)
    for name, v in sorted(c.items()):
        if name.endswith('_dispatches'): continue
        if name in ('FETCH_SIZE', 'WRITE_SIZE'):
            dd = max(c.get(name + '_dispatches', 1.0), 1.0)
            corr = 2.0 if name == 'FETCH_SIZE' else 1.0          # gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md)
            print(f'   {name:28s} {v / dd * 1024 * corr / 1e9:16.3f} GB per dispatch (KiB counter{", x2 gfx950 correction" if corr > 1 else ""})')
            continue
        print(f'   {name:28s} {v / d:16.0f} per dispatch   {100.0 * v / wc:7.2f} % of WAVE_CYCLES')
PY
  find $OUT -name '*.db' -delete; find $OUT -name '*.csv' -size +2M -delete
done
