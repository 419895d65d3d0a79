#!/bin/bash
# one-box sweep of a few launch-geometry fields on the final round-6 build (bench.py --tuning FIELD=V, 20 steps each; base line first and last)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-40s %.3f ms' % ('$1' or 'defaults', d['ms_per_step']))"; }
run ""
for v in 128 192 384 512; do run "--tuning ppgemm_tn_items=$v"; done
for v in 4096 16384; do run "--tuning wgrad_slice_tokens=$v"; done
for v in 256 1024; do run "--tuning wgrad_blocks=$v"; done
for v in 1 2; do run "--tuning lstm_scan3_rb128=$v"; done
run "--tuning lstm_scan3_rb256=2"
run "--tuning route_wgrad_stream=1"
run ""
