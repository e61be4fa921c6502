#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for z in 208 224 240 256 288; do ESN0=-0.5 bash tools/exp_run.sh 1 $z; done
