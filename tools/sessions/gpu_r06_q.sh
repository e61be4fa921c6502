#!/bin/bash
# round 6, session q: cfg4 -- the big-Z buckets on their compile-time kernels, launched FIRST on streams of their own (NRLDPC_MULTI_ORDER=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06q; mkdir -p $O
run() { echo "== $*" | tee -a $O/probe.txt; env "$@" python tools/probe_multi.py 2>&1 | grep -v amdgpu.ids | tail -7 | tee -a $O/probe.txt; }
run A=1
for q in 4 8; do
for rows in 49152 24576 16384 8192 4096; do
  run GPU_MAX_HW_QUEUES=$q NRLDPC_MULTI_STREAMS=7 NRLDPC_MULTI_ORDER=1 NRLDPC_MULTI_Z64_MIN_ROWS=$rows
done
done
run GPU_MAX_HW_QUEUES=8 NRLDPC_MULTI_STREAMS=7 NRLDPC_MULTI_ORDER=0 NRLDPC_MULTI_Z64_MIN_ROWS=16384
run A=1
