#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_lstm_scan.py -x -q -m gpu -k scan3 > gpurun_out/r6c/pytest_scan3.log 2>&1; tail -5 gpurun_out/r6c/pytest_scan3.log
timeout 300 python profiles/microbench_lstm_scan3.py 2>&1 | tee gpurun_out/r6c/microbench_scan3.txt
