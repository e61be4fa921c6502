#!/bin/bash
# round 6, call l: long differential fuzz of the final tree's decoder kernels against the oracle -- the parity pass of every kernel family changed
# this round (row_parity_z64 reads in pinned batches), and the refill counters carry events: new seeds, every mode.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06l; mkdir -p $O; rm -rf $O/*
for m in 0 1 3; do REFILL=1 NRLDPC_REFILL_MASK=$m timeout 1500 python tools/fuzz_decode.py 1500 $((160+m)) > $O/fuzz_refill_mask$m.log 2>&1; tail -1 $O/fuzz_refill_mask$m.log; done
SMALL=1 timeout 2400 python tools/fuzz_decode.py 10000 164 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 2400 python tools/fuzz_decode.py 8000 165 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
AUTO=1 timeout 1500 python tools/fuzz_decode.py 4000 166 > $O/fuzz_auto.log 2>&1; tail -1 $O/fuzz_auto.log
MULTI=1 timeout 1500 python tools/fuzz_decode.py 2000 167 > $O/fuzz_multi.log 2>&1; tail -1 $O/fuzz_multi.log
