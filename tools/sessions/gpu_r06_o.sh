#!/bin/bash
# round 6, session o: the reference's second harness at its own defaults, on the device
mkdir -p gpurun_out/r06o
python tools/demo_snr_vs_a.py > gpurun_out/r06o/snr_vs_a.txt 2>&1
tail -50 gpurun_out/r06o/snr_vs_a.txt
