#!/bin/bash
# round 6, call f: how long should an idle copy thread poll before it sleeps?  (call e: 300 us halves the copy / quantise phase of every dtype.)
# NRLDPC_HOST_SPIN_US is read once per process: one process per value, the values interleaved twice; cgroup throttling counters per run.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06f; mkdir -p $O; rm -rf $O/*
for rep in 1 2; do for sp in 0 100 300 50 200 1000; do
  PROBE_PATHS=2 NRLDPC_HOST_SPIN_US=$sp timeout 300 python tools/probe_quant_prefetch.py 2>&1 | grep "^{" | cut -c1-330 | tee -a $O/spin.txt
done; done
