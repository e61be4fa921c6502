#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/test_decode_gpu.py tests/test_chain_gpu.py tests/test_full_size_gpu.py tests/test_step_golden.py tests/test_testbench_gpu.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/gputests4.log 2>&1; cat gpurun_out/gputests4.log
bash tools/exp_run.sh 1 384
python tools/bench_chain.py > gpurun_out/chain.log 2>&1; grep -E "receive chain|crc" gpurun_out/chain.log | cut -c1-70,120-330
python tools/bench_montecarlo.py 2>&1 | tail -3 | cut -c1-300
