#!/bin/bash
# one bench line per RvtTuning override (Base-1Mpx, 10 timed steps each): prints ms per step.  usage: profiles/sweep_tuning.sh "a=1" "b=2 c=3" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for spec in "$@"; do
  flags=""; for kv in $spec; do [ "$kv" != "none" ] && flags="$flags --tuning $kv"; done
  ms=$(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $flags 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$spec -> $ms ms"
done
