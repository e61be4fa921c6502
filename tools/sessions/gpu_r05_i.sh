#!/bin/bash
# round 5, session i: differential fuzz against the oracle on the final kernels: the refill path (three refill periods), the small sizes, the large ones
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05i; mkdir -p $O; rm -rf $O/*
for m in 0 1 3; do REFILL=1 NRLDPC_REFILL_MASK=$m timeout 900 python tools/fuzz_decode.py 400 $((50+m)) > $O/fuzz_refill_mask$m.log 2>&1; tail -1 $O/fuzz_refill_mask$m.log; done
SMALL=1 timeout 900 python tools/fuzz_decode.py 1500 61 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 900 python tools/fuzz_decode.py 800 62 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
