#!/bin/bash
# ISA audit of one kernel of the built library: disassembles the gfx950 code object of ONE translation unit
# (rvt_amd/csrc/_obj/capi_<part>.o; llvm-objdump --offloading), extracts the kernel whose mangled name starts with <prefix> into
# $ISA_DIR/k.s and prints, for every loop (backward branch), the instruction histogram: VALU / MFMA / LDS / VMEM counts.
# This is how the 8.5-slots-per-GELU and the v_mov / v_alignbit packing overhead of the chain MLP kernels were found.
#   usage: profiles/isa_hist.sh <part: core|conv|linear|stem|mlp|attn|lstm|scan> <kernel-mangled-prefix>
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=${ISA_DIR:-/tmp/isa}
mkdir -p $D; cd $D
cp $ROOT/rvt_amd/csrc/_obj/capi_$1.o part.o
rm -f part.o.*; /opt/rocm/lib/llvm/bin/llvm-objdump --offloading part.o >/dev/null 2>&1
mv part.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 rvt.co; rm -f part.o.*
/opt/rocm/lib/llvm/bin/llvm-objdump -d rvt.co > rvt.s
L=$(grep -n "^[0-9a-f]* <$2" rvt.s | head -1 | cut -d: -f1)
awk -v s=$L 'NR>s && /^[0-9a-f]+ </{exit} NR>=s{print}' rvt.s > k.s
python3 - <<'PY'
import re
from collections import Counter
import os
lines=open(os.path.join(os.environ.get('ISA_DIR','/tmp/isa'),'k.s')).read().split('\n')
base=int(lines[0].split()[0],16)
addr={}
for i,l in enumerate(lines):
    m=re.search(r'// ([0-9A-F]{12}):',l)
    if m: addr[int(m.group(1),16)]=i
print(f'kernel: {lines[0].strip()[:120]}  ({len(lines)} lines)')
for i,l in enumerate(lines):
    m=re.search(r's_cbranch_\w+ \d+ .*\+0x([0-9a-f]+)>',l)
    if m:
        t=base+int(m.group(1),16)
        if t in addr and addr[t]<i:
            c=Counter(x.split()[0] for x in lines[addr[t]:i+1] if x.strip())
            valu=sum(v for k,v in c.items() if k.startswith('v_') and 'mfma' not in k)
            mf=sum(v for k,v in c.items() if 'mfma' in k)
            lds=sum(v for k,v in c.items() if k.startswith('ds_'))
            vm=sum(v for k,v in c.items() if k.startswith(('global_','buffer_','scratch_')))
            wt=sum(v for k,v in c.items() if k.startswith('s_waitcnt'))
            trans=sum(v for k,v in c.items() if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)',k))
            print(f'loop lines {addr[t]}..{i} ({i-addr[t]} instrs): VALU {valu} (transcendental {trans}) MFMA {mf} LDS {lds} VMEM {vm} waitcnt {wt} barriers {c.get("s_barrier",0)} scratch {sum(v for k,v in c.items() if k.startswith("scratch_"))}')
            top=[(k,v) for k,v in c.most_common(14)]
            print('    ', top)
PY
