#!/bin/bash
# round 6, session s: the run-time-Z kernel with the next layer's ring offsets prefetched into SGPRs (new) against the tree before it (old), alternated
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06s; mkdir -p $O; : > $O/ab.txt
OLD=$PWD/exp_libs/lib_gen_old.so; NEW=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip.so
for rep in 1 2; do
for tag in old new; do
  [ $tag = old ] && L=$OLD || L=$NEW
  echo "== $tag cfg4" >> $O/ab.txt
  NRLDPC_LIB=$L python tools/probe_multi.py 2>&1 | grep "all done" | tail -5 >> $O/ab.txt
  for spec in "1 384 4096" "2 384 4096" "1 160 8192" "1 64 16384" "2 20 65536" "1 8 65536"; do
    for et in 0 1; do
      NRLDPC_LIB=$L NRLDPC_FORCE_GENERIC=1 python tools/bench_one.py $spec $et 0 2>&1 | grep Gbit | sed "s/^[^ ]* /$tag /" >> $O/ab.txt
    done
  done
done
done
cat $O/ab.txt
