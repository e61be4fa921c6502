#!/bin/bash
# round-4 GPU session C: BG2 packed row-form A/B (fixed + parity stop), cfg4 launch order A/B, whole GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r04c_pr.log
for z in 88 96 176 352 144 160 288 320; do
  python tools/exp_check.py 2 $z 2>&1 | grep -E "Gbit|FAIL" | sed "s/^/packed-row /" | tee -a gpurun_out/r04c_pr.log
  NRLDPC_NO_PACKED_ROW=1 python tools/exp_check.py 2 $z 2>&1 | grep -E "Gbit|FAIL" | sed "s/^/previous   /" | tee -a gpurun_out/r04c_pr.log
done
for z in 88 352; do for nl in 9 22; do
  python tools/exp_check.py 2 $z $nl 2>&1 | grep -E "Gbit|FAIL" | sed "s/^/packed-row /" | tee -a gpurun_out/r04c_pr.log
  NRLDPC_NO_PACKED_ROW=1 python tools/exp_check.py 2 $z $nl 2>&1 | grep -E "Gbit|FAIL" | sed "s/^/previous   /" | tee -a gpurun_out/r04c_pr.log
done; done
for i in 1 2; do
python tools/bench_configs.py --only-mixed 2>&1 | grep -E "wall_ms" | sed "s/^/sorted /" | cut -c1-330 | tee -a gpurun_out/r04c_cfg4.log
NRLDPC_MULTI_KEEP_ORDER=1 python tools/bench_configs.py --only-mixed 2>&1 | grep -E "wall_ms" | sed "s/^/caller-order /" | cut -c1-330 | tee -a gpurun_out/r04c_cfg4.log
done
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r04c_pytest.log
