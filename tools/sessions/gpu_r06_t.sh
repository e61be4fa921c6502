#!/bin/bash
# round 6, session t: the headline unit (BG1 Z = 384, split form) compiled under alternative LLVM scheduling strategies, against the shipped build; alternated
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06t; mkdir -p $O; : > $O/ab4.txt
D=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip.so
for rep in 1 2 3; do
  for v in default 19 20 21 24; do
    [ $v = default ] && L=$D || L=$PWD/exp_libs/lib_sched$v.so
    for et in 0 1; do
      NRLDPC_LIB=$L python tools/bench_one.py 1 384 4096 $et 0 2>&1 | grep Gbit | sed "s/^[^ ]* /v$v /" >> $O/ab4.txt
    done
  done
done
cat $O/ab4.txt
