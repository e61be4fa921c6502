#!/bin/bash
# round 6, call a: 16-bit datapath ubench + the bench line of the untouched tree on this box
mkdir -p gpurun_out/r06a; cd /root/repo
./tools/ubench/valu_rate3.bin 2>&1 | tee gpurun_out/r06a/valu_rate3.txt
python bench.py --steps 20 --warmup 5 2>gpurun_out/r06a/bench.err | tee gpurun_out/r06a/bench.json | cut -c1-600
