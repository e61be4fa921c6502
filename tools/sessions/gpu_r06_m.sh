#!/bin/bash
# round 6, call m: the Monte-Carlo loop with the payload draw as one library kernel (nrldpc_payload_bits_dev)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06m; mkdir -p $O; rm -rf $O/*
timeout 1200 python -m pytest tests/test_harness_gpu.py tests/test_abi_caller_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
python tools/bench_montecarlo.py 2>&1 | grep "^{" | cut -c1-300 | tee $O/mc.txt
cp gpurun_out/bench_montecarlo.json $O/
