#!/bin/bash
mkdir -p gpurun_out/r05ac; cd /root/repo
python tools/demo_bler_curve.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05ac/bler_curves.txt
