#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes for the headline bench.
# Usage: tools/profile_gpu.sh <tag> [extra bench args]
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-e2e $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
pmc() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $BENCH > $OUT/$name.log 2>&1
}
pmc sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY
pmc sqB SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
find $OUT -name "*.csv" | head -40
