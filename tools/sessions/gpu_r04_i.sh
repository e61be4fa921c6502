#!/bin/bash
# round-4 GPU session I: the whole GPU suite, then the measurement set behind profiles/r04_* (interleaved block geometry in)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/final_session.sh
