#!/bin/bash
# Monte-Carlo loop with several shards on one GPU, each on its own stream
# (the shard runs were a variant of tools/bench_montecarlo.py and harness.simulate_point_device that was reverted after this measurement: profiles/r05_montecarlo_shards_rejected.txt)
mkdir -p gpurun_out/r05r; cd /root/repo
timeout 900 python -m pytest tests/test_harness_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05r/tests.txt
python tools/bench_montecarlo.py 2>&1 | grep "^{" | tee gpurun_out/r05r/mc.txt
python tools/bench_montecarlo.py 2>&1 | grep "^{" | tee gpurun_out/r05r/mc2.txt
