#!/bin/bash
# The measurement set behind profiles/r02_*: host path (+ spread over processes), bench line, every BASELINE configuration, chain stages.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/bench_host_path.py > gpurun_out/hostpath.log 2>&1
SPINS="0 0" bash tools/host_variance.sh > gpurun_out/host_path_variance.txt 2>&1
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_line.json
python tools/bench_configs.py > gpurun_out/cfg.log 2>&1
python tools/bench_chain.py > gpurun_out/chain.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hostpath -o hp -- python $GRAFT_REPO_ROOT/tools/bench_host_path.py --big-only > $GRAFT_REPO_ROOT/gpurun_out/prof_hostpath.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/hostpath.log | cut -c1-200; cat gpurun_out/host_path_variance.txt; cut -c1-300 gpurun_out/bench_line.json
