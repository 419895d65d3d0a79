#!/bin/bash
# round 5: A/B of bench.py under tuning overrides (same box, back to back).  usage: gpu_r5c.sh <tag> "<override> ..." e.g. "route_wgrad_stream=2"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5c}
mkdir -p $OUT; cd $ROOT
shift
python bench.py --steps 15 --warmup 4 --no-cpu-baseline > $OUT/ab_base.json 2> $OUT/ab_base.err
echo "baseline: $(python -c "import json;d=json.load(open('$OUT/ab_base.json'));print(d['ms_per_step'])") ms"
for OV in "$@"; do
  ARGS=""; for kv in $OV; do ARGS="$ARGS --tuning $kv"; done
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline $ARGS > $OUT/ab_${OV// /_}.json 2> $OUT/ab_${OV// /_}.err
  echo "$OV: $(python -c "import json;d=json.load(open('$OUT/ab_${OV// /_}.json'));print(d['ms_per_step'])") ms"
done
python bench.py --steps 15 --warmup 4 --no-cpu-baseline > $OUT/ab_base2.json 2> $OUT/ab_base2.err
echo "baseline again: $(python -c "import json;d=json.load(open('$OUT/ab_base2.json'));print(d['ms_per_step'])") ms"
