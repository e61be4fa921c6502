#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/test_decode_gpu.py tests/test_chain_gpu.py tests/test_full_size_gpu.py tests/test_harness_gpu.py tests/test_step_golden.py tests/test_system_objects_gpu.py tests/test_testbench_gpu.py tests/test_abi_caller_gpu.py tests/test_encode_gpu.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/gputests2.log 2>&1; cat gpurun_out/gputests2.log
python tools/exp_check.py 1 384 2>&1 | grep -E "Gbit|FAIL|rror" | tee gpurun_out/exp_default.log
ESN0=-3.0 python tools/exp_check.py 1 384 2>&1 | grep -E "Gbit|FAIL|rror" | sed "s/^/noconv /" | tee -a gpurun_out/exp_default.log
python tools/exp_check.py 2 384 2>&1 | grep -E "Gbit|FAIL|rror" | tee -a gpurun_out/exp_default.log
python tools/bench_chain.py > gpurun_out/chain.log 2>&1; tail -5 gpurun_out/chain.log | cut -c1-400
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cndmask.hip -o /tmp/valu_cndmask 2>/dev/null && /tmp/valu_cndmask | tee gpurun_out/ubench_cndmask.txt
