cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; OUT=gpurun_out/exp_pg.log; : > $OUT
run() { python tools/exp_check.py $1 $2 $3 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT
  NRLDPC_NO_PACKED_GENERAL=1 python tools/exp_check.py $1 $2 $3 2>&1 | grep -E "Gbit|FAIL|Error|error" | sed 's/^default/runtimeZ/' | tee -a $OUT; }
run 2 20 8; run 2 8 20; run 2 32 12; run 2 56 30; run 2 80 16; run 1 16 10; run 1 32 24; run 1 48 6; run 1 80 30; run 1 6 40
