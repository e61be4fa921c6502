#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05j; mkdir -p $O; rm -rf $O/*
for rep in 1 2; do
timeout 600 python tools/exp_row_refill.py 2,384 2,144 1,320 2,192 2,288 >> $O/default.txt 2>&1
NRLDPC_LIB=$PWD/exp_libs/lib_rowrefill.so timeout 600 python tools/exp_row_refill.py 2,384 2,144 1,320 2,192 2,288 >> $O/rowrefill.txt 2>&1
done
grep -h "^{\|PARITY\|Error" $O/default.txt $O/rowrefill.txt
