cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q1
python -m pytest tests -m gpu -q -x -k "pack or weights or host or production or backbone" 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --op-breakdown gpurun_out/q1/op.txt > gpurun_out/q1/bench.json 2> gpurun_out/q1/bench.err
grep -h "rvt_pack_table\|sum =" gpurun_out/q1/op.txt | head -3
