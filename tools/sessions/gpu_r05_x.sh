#!/bin/bash
# round 5: the long fuzz (new seeds, ten times the usual counts)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05x; mkdir -p $O; rm -rf $O/*
for m in 0 1 3; do REFILL=1 NRLDPC_REFILL_MASK=$m timeout 1500 python tools/fuzz_decode.py 1500 $((80+m)) > $O/fuzz_refill_mask$m.log 2>&1; tail -1 $O/fuzz_refill_mask$m.log; done
SMALL=1 timeout 2400 python tools/fuzz_decode.py 10000 84 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 2400 python tools/fuzz_decode.py 8000 85 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
AUTO=1 timeout 1500 python tools/fuzz_decode.py 4000 86 > $O/fuzz_auto.log 2>&1; tail -1 $O/fuzz_auto.log
MULTI=1 timeout 1500 python tools/fuzz_decode.py 2000 87 > $O/fuzz_multi.log 2>&1; tail -1 $O/fuzz_multi.log
