cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; OUT=gpurun_out/exp_pk2.log; : > $OUT
for bg in 1 2; do for z in 36 40 44 48 56; do
  python tools/exp_check.py $bg $z 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT
  NRLDPC_FORCE_GENERIC=1 python tools/exp_check.py $bg $z 2>&1 | grep -E "Gbit|FAIL|Error|error" | sed 's/^default/generic/' | tee -a $OUT
  l=exp_libs/lib_${bg}_${z}_0_0_rw3.so
  [ -f $l ] && NRLDPC_LIB=$PWD/$l python tools/exp_check.py $bg $z 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT
done; done
