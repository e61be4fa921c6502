#!/bin/bash
# round-4 GPU session G: decoder tests, BASELINE configurations (the parity-stop regression check), then the measurement set again
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04g_pytest.log
SKIP_TESTS=1 bash tools/final_session.sh
