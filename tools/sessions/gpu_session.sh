#!/bin/bash
# What a GPU session of this repository usually runs (via gpurun): the GPU test suite, then the headline bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py --steps 10 --warmup 2 2>/dev/null | tail -c 400
