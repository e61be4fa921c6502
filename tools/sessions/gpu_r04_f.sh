#!/bin/bash
# round-4 GPU session F: the whole GPU suite on the final build, then the bench line (profile of this build is committed)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 3400 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) > gpurun_out/gputests.log 2>&1; cat gpurun_out/gputests.log | cut -c1-300
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_line.json
cut -c1-300 gpurun_out/bench_line.json
