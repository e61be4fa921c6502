#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06n; mkdir -p $O; rm -rf $O/*
for seq in f32 f32 f16,f32 f16,f32 f16,f32,f64,f32 f32,f32,f32,f32 f16+,f32 f16,f16,f32 f64,f32; do
  echo "== $seq"; timeout 120 python tools/probe_host_leg_order.py $seq 2>/dev/null | grep "^{"
done | tee $O/order.txt
for q in 2 8; do echo "== GPU_MAX_HW_QUEUES=$q f16,f32,f64,f32"; GPU_MAX_HW_QUEUES=$q timeout 120 python tools/probe_host_leg_order.py f16,f32,f64,f32 2>/dev/null | grep "^{"; done | tee -a $O/order.txt
