#!/bin/bash
# GPU side of a kernel experiment round: every exp_libs/lib_<bg>_<z>[_nlN]_*.so and the default library through tools/exp_check.py.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/exp_$(date +%H%M%S).log
for spec in ${SPECS:-"1 384"}; do :; done
run() { # bg z [nl]
  python tools/exp_check.py $1 $2 $3 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT
  for l in exp_libs/lib_$1_$2${3:+_nl$3}_*.so; do
    [ -f "$l" ] && NRLDPC_LIB=$PWD/$l python tools/exp_check.py $1 $2 $3 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT
  done
}
if [ -n "$1" ]; then run "$@"; else run 1 384; fi
