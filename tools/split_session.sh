#!/bin/bash
# GPU session: the split kernels forced on (NRLDPC_SPLIT=1) through the whole GPU test suite, then both forms timed on
# every (BG, Z) and every BASELINE configuration.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time NRLDPC_SPLIT=1 timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/gputests_split1.log 2>&1; cat gpurun_out/gputests_split1.log
for s in 0 1; do
  NRLDPC_SPLIT=$s OUT_SUFFIX=_split$s python tools/bench_all_z.py > gpurun_out/allz_split$s.log 2>&1
  NRLDPC_SPLIT=$s OUT_SUFFIX=_split$s python tools/bench_configs.py > gpurun_out/cfg_split$s.log 2>&1
done
tail -3 gpurun_out/allz_split1.log gpurun_out/cfg_split1.log
