#!/usr/bin/env python3
"""Instruction histogram / scratch-access locations of ONE kernel in a -save-temps .s file.
usage: isa_kernel.py file.s <mangled-name-substring> [--dump]"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
body = s[start:end + 1]
print(s[start], len(body), 'lines')
cnt = collections.Counter()
for i, l in enumerate(body):
    t = l.strip().split()
    if t and not t[0].startswith(('.', ';')) and not t[0].endswith(':'):
        cnt[t[0]] += 1
for k, v in cnt.most_common(60):
    print(f'{v:6d} {k}')
if '--dump' in sys.argv:
    open('/tmp/isa/kernel_body.s', 'w').write('\n'.join(body))
