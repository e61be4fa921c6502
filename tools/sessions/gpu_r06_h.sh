#!/bin/bash
# round 6, call h: the RCCL-success branch of bench.py's Comm with one rank (BENCH_COMM_WORLD1)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06h
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "probed_rccl or survives" 2>&1 | tail -25 | tee gpurun_out/r06h/tests.txt
