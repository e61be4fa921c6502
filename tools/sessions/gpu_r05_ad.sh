#!/bin/bash
mkdir -p gpurun_out/r05ad; cd /root/repo
timeout 900 python -m pytest tests/test_harness_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05ad/tests.txt
python tools/demo_bler_curve.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05ad/bler_curves.txt
