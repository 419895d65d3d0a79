cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q3
python profiles/microbench_pack.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/q3/pack.txt
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --op-breakdown gpurun_out/q3/op.txt > gpurun_out/q3/bench.json 2> gpurun_out/q3/bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/q3/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])"
