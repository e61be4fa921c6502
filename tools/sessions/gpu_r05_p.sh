#!/bin/bash
# differential fuzz of nrldpc_decode_multi_dev (shared launches by workgroup class, routing, per-handle layer counts) against the oracle
mkdir -p gpurun_out/r05p; cd /root/repo
MULTI=1 timeout 1500 python tools/fuzz_decode.py 300 11 > gpurun_out/r05p/fuzz_multi.log 2>&1; tail -3 gpurun_out/r05p/fuzz_multi.log
MULTI=1 NRLDPC_MULTI_Z64_MIN_ROWS=2048 timeout 900 python tools/fuzz_decode.py 120 12 > gpurun_out/r05p/fuzz_multi_routed.log 2>&1; tail -3 gpurun_out/r05p/fuzz_multi_routed.log
MULTI=1 NRLDPC_MULTI_CLASS_EDGES=128,192,320,448 timeout 900 python tools/fuzz_decode.py 120 13 > gpurun_out/r05p/fuzz_multi_edges.log 2>&1; tail -3 gpurun_out/r05p/fuzz_multi_edges.log
