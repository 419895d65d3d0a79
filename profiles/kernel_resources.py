"""Compact per-kernel register / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python profiles/kernel_resources.py [filter-substring]"""
import re, subprocess, sys

def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n

cur, rows = None, []
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = dict(name=m.group(1)); rows.append(cur); continue
    for key, pat in (('v', r' VGPRs: (\d+)'), ('a', r'AGPRs: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
                     ('scr', r'ScratchSize \[bytes/lane\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for r in rows:
    d = demangle(r['name'])
    if d.startswith('_ZN3rvt'):
        d = re.sub(r'^_ZN3rvt\d+', '', d).replace('DF16b', 'bf16').replace('NS_', ' ').replace('Lb0E', ' 0').replace('Lb1E', ' 1')
        d = re.sub(r'EEEEvT3_.*$', '', d)
    d = re.sub(r'\(.*$', '', d).replace('rvt::', '').replace('__bf16', 'bf16')
    d = d.replace('void ', '')
    if flt in d:
        print(f"{r.get('v',0):4d}v {r.get('a',0):4d}a occ={r.get('occ',0)} scratch={r.get('scr',0):4d} lds={r.get('lds',0):6d}  {d[:150]}")
