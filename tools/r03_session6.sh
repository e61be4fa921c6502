#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/exp_run.sh 1 384
ESN0=7.5 bash tools/exp_run.sh 1 384 5
ESN0=3.0 bash tools/exp_run.sh 1 384 13
ESN0=1.0 bash tools/exp_run.sh 1 384 24
