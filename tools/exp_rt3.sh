#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/exp_rt3.log
: > $OUT
L=$PWD/exp_libs
for rep in 1 2 3; do
for et in 0 1; do
  NRLDPC_LIB=$L/lib_1_384_2_3_base.so python tools/bench_one.py 1 384 4096 $et 24 2>&1 | grep Gbit | sed "s/^/listed /" | tee -a $OUT
  for v in alllate late24 both exit24 exit24both; do
    NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_LIB=$L/lib_1_384_2_3_$v.so python tools/bench_one.py 1 384 4096 $et 24 2>&1 | grep Gbit | sed "s/^/rt /" | tee -a $OUT
  done
done
done
