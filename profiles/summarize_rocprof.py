"""Aggregate rocprofv3 CSV output (kernel stats + per-dispatch PMC rows) into a small text summary."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r'\(.*$', '', name)
    m = re.search(r'rvt::(\w+)<(.*)', name)
    if not m:
        return name[:100]
    kern, rest = m.group(1), m.group(2)
    tags = []
    for t in ('Im2colSrc', 'DgradSrc', 'ConcatSrc', 'XfGelu', 'EpStore', 'EpScaleRes', 'EpGeluBwd', 'EpSplit2',
              'EpAtomicF32', 'EpDgradScatter', 'EpLstm'):
        if t in rest:
            tags.append(t)
    head = rest.split(',')[:3]
    return f'{kern}<{",".join(h.strip() for h in head)}|{"+".join(tags)}>'


for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    print(f'== kernel stats ({os.path.relpath(f, out)}) ==')
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        print(f"{short(r['Name']):90s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs']) / 1e6:10.3f} "
              f"avg_us={float(r['AverageNs']) / 1e3:10.2f} pct={r['Percentage']}")

for what in ('fetch', 'write'):
    files = glob.glob(os.path.join(out, what, '**', '*counter_collection.csv'), recursive=True)
    for f in files:
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value'])
        print(f'== {what.upper()}_SIZE per kernel ({os.path.relpath(f, out)}); counter unit = KiB, raw (uncorrected) ==')
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f'{k:90s} dispatches={n:6d} sum={v / 1024 / 1024:10.3f} GiB avg={v / n / 1024:10.3f} MiB')


# ---- per-entry-point HBM traffic (bytes per launch) for bench.py's `roofline.traffic` --------------------------------
# gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports wide coalesced reads by exactly 2x; WRITE_SIZE was
# calibrated here on ln_fwd (equal read and write bytes) and matches the byte count 1:1.  Counter unit: KiB.
import json

ENTRY = {   # kernel-name predicate -> C entry point
    'rvt_linear_wgrad': lambda k: 'gemm_kernel' in k and 'Lb1ELb0' in k and 'PlainSrc' in k and 'ConcatSrc' not in k and 'Im2colSrc' not in k,
    'rvt_linear_dgrad': lambda k: False,
}
raw = {}
for what in ('fetch', 'write'):
    for f in glob.glob(os.path.join(out, what, '**', '*counter_collection.csv'), recursive=True):
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            for ep, pred in ENTRY.items():
                if pred(r['Kernel_Name']):
                    e = raw.setdefault(ep, {'fetch': 0.0, 'write': 0.0, 'n_fetch': 0, 'n_write': 0})
                    e[what] += float(r['Counter_Value'])
                    e['n_' + what] += 1
        # round 3: the stage-3 / stage-4 weight gradients run on ppgemm_tn_kernel, which rvt_lstm_wgrad launches too.  Per
        # stage the backward issues [lstm_wgrad, then fc2 / fc1 / proj / qkv weight gradients of the two blocks]: of every nine
        # consecutive ppgemm_tn dispatches the first is the ConvLSTM's (checked: it is the largest of its group, its rows
        # being 4C + 2C wide) and is left out of the rvt_linear_wgrad average.
        tn = sorted((r for r in rows if 'ppgemm_tn_kernel' in r['Kernel_Name']), key=lambda r: int(r['Dispatch_Id']))
        if tn and len(tn) % 9 == 0:
            e = raw.setdefault('rvt_linear_wgrad', {'fetch': 0.0, 'write': 0.0, 'n_fetch': 0, 'n_write': 0})
            for g0 in range(0, len(tn), 9):
                grp = [float(r['Counter_Value']) for r in tn[g0:g0 + 9]]
                if what == 'fetch':
                    assert grp[0] == max(grp), 'ppgemm_tn dispatch order changed: the ConvLSTM launch is not first of its group'
                e[what] += sum(grp[1:])
                e['n_' + what] += 8
res = {}
for ep, e in raw.items():
    if e['n_fetch'] and e['n_write']:
        res[ep] = {'launches_profiled': e['n_fetch'],
                   'fetch_size_kib_raw_per_launch': e['fetch'] / e['n_fetch'],
                   'write_size_kib_raw_per_launch': e['write'] / e['n_write'],
                   'traffic_bytes_per_launch': int(1024 * (2 * e['fetch'] / e['n_fetch'] + e['write'] / e['n_write']))}
json.dump(res, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
print('== traffic per launch ==', json.dumps(res))
