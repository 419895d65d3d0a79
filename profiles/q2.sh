cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q2
python profiles/microbench_pack.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/q2/pack.txt
python -m pytest tests -m gpu -q -x -k "attn_block or stage_driver or production or backbone or step" 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --tuning route_attn_preln=$v --op-breakdown gpurun_out/q2/op_$v.txt > gpurun_out/q2/bench_${v}_$rep.json 2> gpurun_out/q2/bench_${v}_$rep.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/q2/bench_${v}_$rep.json') if l.startswith('{')][-1]); print('preln=$v', $rep, d['ms_per_step'], d['value'])"
done; done 2>&1 | tee gpurun_out/q2/ab.txt
grep -h "attn_block_bwd\|layernorm_bwd " gpurun_out/q2/op_0.txt gpurun_out/q2/op_1.txt
