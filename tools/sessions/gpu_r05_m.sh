#!/bin/bash
# cfg4 (one nrldpc_decode_multi_dev call over 102 (BG, Z) buckets): how many of the buckets get a launch of their own on the
# compile-time-Z kernels, over how many streams / hardware queues
mkdir -p gpurun_out/r05m; cd /root/repo
O=gpurun_out/r05m/multi_streams.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/probe_multi.py 2>&1 | grep -v "^largest\|^mean" | tail -4 >> $O; }
run A=0
run NRLDPC_MULTI_Z64_MIN_ROWS=1
run NRLDPC_MULTI_Z64_MIN_ROWS=1 NRLDPC_MULTI_STREAMS=7 GPU_MAX_HW_QUEUES=8
run NRLDPC_MULTI_Z64_MIN_ROWS=1 NRLDPC_MULTI_STREAMS=15 GPU_MAX_HW_QUEUES=16
run NRLDPC_MULTI_Z64_MIN_ROWS=1 NRLDPC_MULTI_STREAMS=31 GPU_MAX_HW_QUEUES=32
run NRLDPC_MULTI_Z64_MIN_ROWS=1 NRLDPC_MULTI_STREAMS=31
run NRLDPC_MULTI_Z64_MIN_ROWS=12288 NRLDPC_MULTI_STREAMS=15 GPU_MAX_HW_QUEUES=16
run NRLDPC_MULTI_Z64_MIN_ROWS=1 NRLDPC_MULTI_STREAMS=15 GPU_MAX_HW_QUEUES=16 NRLDPC_NO_REFILL=1
run NRLDPC_MULTI_STREAMS=7 GPU_MAX_HW_QUEUES=8
cat $O
