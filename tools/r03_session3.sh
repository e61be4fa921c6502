#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/test_decode_gpu.py tests/test_chain_gpu.py tests/test_full_size_gpu.py tests/test_harness_gpu.py tests/test_step_golden.py tests/test_system_objects_gpu.py tests/test_testbench_gpu.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/gputests3.log 2>&1; cat gpurun_out/gputests3.log
python tools/bench_chain.py > gpurun_out/chain.log 2>&1; grep "receive chain" gpurun_out/chain.log | cut -c1-60,150-420
python tools/bench_small_z.py 2>&1 | grep Gbit > gpurun_out/smallz_new.log
NRLDPC_LIB=$PWD/exp_libs/lib_generic_old.so python tools/bench_small_z.py 2>&1 | grep Gbit > gpurun_out/smallz_old.log
paste gpurun_out/smallz_old.log gpurun_out/smallz_new.log | awk '{print $2,$3,$4,$5,$7, "->", $13,$15}'
