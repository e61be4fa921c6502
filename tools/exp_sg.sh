cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; OUT=gpurun_out/exp_pk4.log; : > $OUT
run() { python tools/exp_check.py $1 $2 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT
  NRLDPC_NO_PACKED=1 python tools/exp_check.py $1 $2 2>&1 | grep -E "Gbit|FAIL|Error|error" | sed 's/^default/nopack /' | tee -a $OUT; }
for z in 36 44 48; do run 1 $z; done
for z in 36 44 48 52 56 72 80; do run 2 $z; done
