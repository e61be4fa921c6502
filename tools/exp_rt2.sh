#!/bin/bash
# GPU side: row-0 lateness variants of the run-time-layer-count build (BG1 Z=384) + the no-barrier timing build of the headline
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/exp_rt2.log
: > $OUT
L=$PWD/exp_libs
for rep in 1 2; do
for nl in 5 24; do for et in 0 1; do
  NRLDPC_LIB=$L/lib_1_384_2_3_base.so python tools/bench_one.py 1 384 4096 $et $nl 2>&1 | grep Gbit | sed "s/^/listed /" | tee -a $OUT
  for v in base alllate late$nl; do
    NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_LIB=$L/lib_1_384_2_3_$v.so python tools/bench_one.py 1 384 4096 $et $nl 2>&1 | grep Gbit | sed "s/^/rt /" | tee -a $OUT
  done
done; done
for v in base nobar; do
  NRLDPC_LIB=$L/lib_1_384_2_3_$v.so python tools/bench_one.py 1 384 4096 0 0 2>&1 | grep Gbit | sed "s/^/headline /" | tee -a $OUT
done
done
