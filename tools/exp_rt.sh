#!/bin/bash
# GPU side of the run-time-layer-count experiment: parity + timing of one (BG, Z) over layer counts, three routes:
#   listed  = compile-time pruned build where one exists (else the RT build), rt = NRLDPC_NO_PRUNED_PIPELINE=1 (RT build), gen = NRLDPC_NO_RT=1 + NRLDPC_NO_PRUNED_PIPELINE=1 (general kernel)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
BG=$1; Z=$2; shift 2
OUT=gpurun_out/exp_rt_${BG}_${Z}.log
: > $OUT
for nl in "$@"; do
  python tools/exp_check.py $BG $Z $nl 2>&1 | grep -E "Gbit|FAIL|Error|error" | sed "s/^/listed /" | tee -a $OUT
  NRLDPC_NO_PRUNED_PIPELINE=1 python tools/exp_check.py $BG $Z $nl 2>&1 | grep -E "Gbit|FAIL|Error|error" | sed "s/^/rt     /" | tee -a $OUT
  NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_NO_RT=1 python tools/exp_check.py $BG $Z $nl 2>&1 | grep -E "Gbit|FAIL|Error|error" | sed "s/^/gen    /" | tee -a $OUT
done
