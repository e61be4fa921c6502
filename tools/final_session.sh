#!/bin/bash
# The measurement set behind profiles/r02_*: GPU tests, host path, bench line, every BASELINE configuration, all lifting sizes,
# chain stages (+ their kernel trace), Monte-Carlo loop, rocprofv3 kernel trace + PMC passes of the bench command.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/gputests.log 2>&1; cat gpurun_out/gputests.log
python tools/bench_host_path.py > gpurun_out/hostpath.log 2>&1
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_line.json
python tools/bench_configs.py > gpurun_out/cfg.log 2>&1
python tools/bench_chain.py > gpurun_out/chain.log 2>&1
python tools/bench_montecarlo.py > gpurun_out/mc.log 2>&1
if [ -z "$SKIP_PROFILES" ]; then
  python tools/bench_all_z.py > gpurun_out/allz.log 2>&1
  bash tools/profile_gpu.sh r02 > gpurun_out/profile.log 2>&1
fi
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_chain -o chain -- python $GRAFT_REPO_ROOT/tools/bench_chain.py > $GRAFT_REPO_ROOT/gpurun_out/prof_chain.log 2>&1 )
cut -c1-300 gpurun_out/bench_line.json
