#!/bin/bash
# round-4 GPU session D: the whole GPU suite (no -x), then the measurement set
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 3400 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/gputests.log 2>&1; cat gpurun_out/gputests.log | cut -c1-300
SKIP_TESTS=1 bash tools/final_session.sh
