#!/bin/bash
# round 6, session u: the whole library compiled with LLVM's post-RA machine scheduler off (-mllvm -enable-post-misched=false; NRLDPC_BUILD_EXTRA_FLAGS, libnrldpc_hip_x.so)
# against the shipped build: every lifting size, 25 fixed iterations and the parity stop at the waterfall; order x, default, default, x
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06u; mkdir -p $O
D=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip.so; X=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip_x.so
i=0
for tag in x default default x; do
  i=$((i+1)); [ $tag = x ] && L=$X || L=$D
  NRLDPC_LIB=$L OUT_SUFFIX=_${tag}$i python tools/bench_all_z.py > $O/allz_${tag}$i.log 2>&1
  NRLDPC_LIB=$L STOP=1 OUT_SUFFIX=_${tag}$i python tools/bench_all_z.py > $O/allz_stop_${tag}$i.log 2>&1
done
cp gpurun_out/bench_all_z*_x?.json gpurun_out/bench_all_z*_default?.json $O/ 2>/dev/null
ls $O
