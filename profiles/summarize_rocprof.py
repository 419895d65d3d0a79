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
