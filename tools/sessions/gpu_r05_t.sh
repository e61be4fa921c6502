#!/bin/bash
# BLER 1e-3 (the reference sweep's stopping point) on 16384 blocks: GPU decoder against the committed sum-product outcomes
mkdir -p gpurun_out/r05t; cd /root/repo
( time timeout 1500 python -m pytest tests/test_bler_gap_gpu.py -m gpu -x -q -s -k "1e3 or (1e2 and cfg5)" 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r05t/bler_1e3.txt
cp gpurun_out/bler_gap.json gpurun_out/r05t/bler_gap_1e3.json
