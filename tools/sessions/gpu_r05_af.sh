#!/bin/bash
mkdir -p gpurun_out/r05af; cd /root/repo
timeout 1500 python -m pytest tests/test_chain_gpu.py tests/test_harness_gpu.py tests/test_system_objects_gpu.py tests/test_testbench_gpu.py tests/test_layers_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05af/tests.txt
bash tools/gpu_r05_ae.sh 2>&1 | head -12
python tools/bench_montecarlo.py 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print(r['config'], round(r['ms_median'],3), round(r['transport_blocks_per_s']/1e6,3))"
