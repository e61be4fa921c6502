#!/bin/bash
# round-4 GPU session E: decoder tests on the merged-group run-time-layer-count builds, then their timing (rt route) on the
# one-thread-per-row sizes and BG1 Z = 384
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04e_pytest.log
export NL_ZS=144,192,320,384 NL_NLS=5,13,17,24,30
python tools/bench_nl.py default > gpurun_out/r04e_nl_default.log 2>&1; cp gpurun_out/bench_nl_default.json gpurun_out/r04e_nl_default.json
NRLDPC_NO_PRUNED_PIPELINE=1 python tools/bench_nl.py rt > gpurun_out/r04e_nl_rt.log 2>&1; cp gpurun_out/bench_nl_rt.json gpurun_out/r04e_nl_rt.json
grep -E "Z=384|Z=320" gpurun_out/r04e_nl_rt.log | head -60
