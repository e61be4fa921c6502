cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; OUT=gpurun_out/exp_pk.log; : > $OUT
for bg in 1 2; do for z in 3 8 16 22 24 26 32; do
  python tools/exp_check.py $bg $z 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT
  NRLDPC_FORCE_GENERIC=1 python tools/exp_check.py $bg $z 2>&1 | grep -E "Gbit|FAIL|Error|error" | sed 's/^default/generic/' | tee -a $OUT
done; done
