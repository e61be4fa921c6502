#!/bin/bash
# round 6, session w: the workgroup-level claim queue of the slot refill (new) against the library before it (old): parity tests of the refill path, then the small sizes of the
# stop landscape, order old new new old
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06w; mkdir -p $O
OLD=$PWD/exp_libs/lib_noqueue.so; NEW=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip.so
( python -m pytest tests/test_refill_gpu.py -x -q 2>&1 | tail -3 ) > $O/refill_tests.txt 2>&1; cat $O/refill_tests.txt
( REFILL=1 NRLDPC_TEST_HOOKS=1 python tools/fuzz_decode.py 600 77 2>&1 | tail -1 ) > $O/fuzz.txt; cat $O/fuzz.txt
export ALLZ_ONLY=2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,26,28,30,32,36,40,44,48,52,56,60,64,72,80,96,128
i=0
for tag in old new new old; do
  i=$((i+1)); [ $tag = old ] && L=$OLD || L=$NEW
  NRLDPC_LIB=$L STOP=1 OUT_SUFFIX=_${tag}$i python tools/bench_all_z.py > $O/stop_${tag}$i.log 2>&1
done
NRLDPC_REFILL_MASK=0 NRLDPC_LIB=$NEW STOP=1 OUT_SUFFIX=_newmask0 python tools/bench_all_z.py > $O/stop_newmask0.log 2>&1
cp gpurun_out/bench_all_z_stop_old?.json gpurun_out/bench_all_z_stop_new?.json gpurun_out/bench_all_z_stop_newmask0.json $O/
ls $O
