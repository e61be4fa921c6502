#!/bin/bash
# round 6, session r: second long differential fuzz of the final tree (new seeds; twice the cases of session l)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06r; mkdir -p $O
export NRLDPC_TEST_HOOKS=1
( for m in 0 1 3; do REFILL=1 NRLDPC_REFILL_MASK=$m python tools/fuzz_decode.py 3000 $((900 + m)) 2>&1 | tail -1 | sed "s/^/refill_mask$m: /"; done ) > $O/refill.txt &
( SMALL=1 python tools/fuzz_decode.py 20000 911 2>&1 | tail -1 | sed "s/^/small: /" ) > $O/small.txt &
( python tools/fuzz_decode.py 16000 912 2>&1 | tail -1 | sed "s/^/large: /" ) > $O/large.txt &
( AUTO=1 python tools/fuzz_decode.py 8000 913 2>&1 | tail -1 | sed "s/^/auto: /" ) > $O/auto.txt &
( MULTI=1 python tools/fuzz_decode.py 4000 914 2>&1 | tail -1 | sed "s/^/multi: /" ) > $O/multi.txt &
wait
cat $O/*.txt
