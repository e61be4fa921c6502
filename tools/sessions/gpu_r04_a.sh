#!/bin/bash
# round-4 GPU session A: decoder GPU tests, all-Z landscape with and without the packed geometry, short bench with host-path trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04a_pytest.log
python tools/bench_all_z.py > gpurun_out/r04a_all_z.log 2>&1
NRLDPC_NO_PACKED=1 OUT_SUFFIX=_nopacked python tools/bench_all_z.py > gpurun_out/r04a_all_z_nopacked.log 2>&1
NRLDPC_HOST_TRACE=1 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
tail -c 3000 gpurun_out/r04a_bench.json
grep "host path" gpurun_out/r04a_bench.err | tail -12
