#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/exp_run.sh 1 384
python tools/bench_montecarlo.py 2>&1 | tail -3 | cut -c1-260
( timeout 900 python -m pytest tests/test_harness_gpu.py -x -q -m gpu 2>&1 | tail -3 )
