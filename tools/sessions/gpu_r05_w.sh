#!/bin/bash
# round 5, last session: every fuzz mode once more on the library as committed (new seeds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05w; mkdir -p $O; rm -rf $O/*
REFILL=1 timeout 900 python tools/fuzz_decode.py 500 71 > $O/fuzz_refill.log 2>&1; tail -1 $O/fuzz_refill.log
SMALL=1 timeout 1200 python tools/fuzz_decode.py 1500 72 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 1200 python tools/fuzz_decode.py 1000 73 > $O/fuzz_large.log 2>&1; tail -1 $O/fuzz_large.log
AUTO=1 timeout 900 python tools/fuzz_decode.py 400 74 > $O/fuzz_auto.log 2>&1; tail -1 $O/fuzz_auto.log
MULTI=1 timeout 900 python tools/fuzz_decode.py 300 75 > $O/fuzz_multi.log 2>&1; tail -1 $O/fuzz_multi.log
