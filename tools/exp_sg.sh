cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; OUT=gpurun_out/exp_sg2.log; : > $OUT
one() { python tools/exp_check.py $1 $2 $3 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT; NRLDPC_LIB=$PWD/$4 python tools/exp_check.py $1 $2 $3 2>&1 | grep -E "Gbit|FAIL|Error|error" | tee -a $OUT; }
for i in 1 2 3; do one 1 384 "" exp_libs/lib_1_384_2_3_sg0.so; done
for i in 1 2; do one 1 384 24 exp_libs/lib_1_384_nl24_2_3_sg0.so; done
for z in 240 224 60 52; do one 2 $z "" exp_libs/lib_2_${z}_2_3_sg1.so; done
