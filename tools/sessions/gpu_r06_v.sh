#!/bin/bash
# round 6, session v: both decoder forms (one / two threads per check row) for every compile-time-Z size, rebuilt with the post-RA scheduler off (NRLDPC_BUILD_AB=1),
# timed again: does the per-size choice of round 3 (z64_split_default) still hold under the new schedule?  order 0, 1, 1, 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06v; mkdir -p $O
export NRLDPC_LIB=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip_ab.so
export ALLZ_ONLY=52,60,64,88,96,104,112,120,128,144,176,192,208,224,240,256,288,320,352,384
i=0
for f in 0 1 1 0; do
  i=$((i+1))
  NRLDPC_SPLIT=$f OUT_SUFFIX=_form${f}_$i python tools/bench_all_z.py > $O/allz_form${f}_$i.log 2>&1
  NRLDPC_SPLIT=$f STOP=1 OUT_SUFFIX=_form${f}_$i python tools/bench_all_z.py > $O/allz_stop_form${f}_$i.log 2>&1
done
cp gpurun_out/bench_all_z*_form?_?.json $O/
ls $O | head -30
