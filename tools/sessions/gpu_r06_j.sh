#!/bin/bash
# round 6, call j: the encoder with a workgroup (four waves) per codeword for Z >= 128: parity tests, then A/B against one wave per codeword
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06j; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_encode_gpu.py tests/test_chain_gpu.py tests/test_testbench_gpu.py tests/test_harness_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
NRLDPC_ENC_NWC=1 timeout 900 python -m pytest tests/test_encode_gpu.py -m gpu -x -q 2>&1 | tail -2 | tee -a $O/tests.txt
NRLDPC_ENC_NWC=4 timeout 900 python -m pytest tests/test_encode_gpu.py -m gpu -x -q 2>&1 | tail -2 | tee -a $O/tests.txt
for v in 4 1 4 1; do NRLDPC_ENC_NWC=$v OUT_SUFFIX=_enc$v python tools/bench_chain.py 2>&1 | grep "'stage': 'encode'" | sed "s/^/nwc=$v /" | cut -c1-250; done | tee $O/encode_ab.txt
python tools/bench_chain.py 2>&1 | grep "'stage': 'encode'" | sed "s/^/default /" | cut -c1-250 | tee -a $O/encode_ab.txt
python tools/bench_montecarlo.py 2>&1 | grep "^{" | cut -c1-250 | tee $O/mc.txt
