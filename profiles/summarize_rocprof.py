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


# ---- per-kernel HBM traffic (bytes per launch) for bench.py's `roofline.traffic` ------------------------------------------
# gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE under-reports wide coalesced reads by exactly 2x;
# WRITE_SIZE was calibrated here on ln_fwd (equal read and write bytes) and matches the byte count 1:1.  Counter unit: KiB.
# Round 4: keyed by the WORKLOAD the profiled command ran (argv[2], e.g. "base_1mpx:bf16:B24:T21") and by kernel instantiation
# (the mangled name as rocprofv3 prints it, truncated), so that bench.py only quotes a traffic figure that belongs to the
# workload it is timing and to the kernel it names - anything else is reported as null.
import json

workload = sys.argv[2] if len(sys.argv) > 2 else 'base_1mpx:bf16:B24:T21'
per = {}
for what in ('fetch', 'write'):
    for f in glob.glob(os.path.join(out, what, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            e = per.setdefault(r['Kernel_Name'][:200], {'fetch': 0.0, 'write': 0.0, 'n_fetch': 0, 'n_write': 0})
            e[what] += float(r['Counter_Value'])
            e['n_' + what] += 1
res, total = {}, 0.0
for k, e in per.items():
    if e['n_fetch'] and e['n_write']:
        b = 1024 * (2 * e['fetch'] / e['n_fetch'] + e['write'] / e['n_write'])
        res[k] = {'launches_profiled': e['n_fetch'], 'fetch_size_kib_raw_per_launch': round(e['fetch'] / e['n_fetch'], 1),
                  'write_size_kib_raw_per_launch': round(e['write'] / e['n_write'], 1), 'traffic_bytes_per_launch': int(b)}
        total += 1024 * (2 * e['fetch'] + e['write'])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7       # bench steps inside the profiled command (warmup + probe + timed)
doc = {workload: {'kernels': res, 'steps_profiled': steps, 'traffic_bytes_per_step': int(total / max(steps, 1))}}
json.dump(doc, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
print(f'== HBM traffic, {workload}: {total / max(steps, 1) / 1e9:.1f} GB per step over {steps} profiled steps, {len(res)} kernel instantiations ==')
