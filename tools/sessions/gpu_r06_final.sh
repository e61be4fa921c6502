#!/bin/bash
# round 6: the whole measurement set on the settled kernels (tools/final_session.sh, TAG=r06), then the counters of the headline code's
# parity-stop launch next to the fixed one and of the stage kernels
cd /root/repo
TAG=r06 bash tools/final_session.sh 2>&1 | tail -20
bash tools/sessions/gpu_r06_stop_pmc.sh 2>&1 | tail -12
bash tools/sessions/gpu_r06_stage_pmc.sh 2>&1 | tail -5
