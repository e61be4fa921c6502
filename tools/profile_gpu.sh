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
BENCH="python $ROOT/bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-e2e --no-early-term $*"
# which build the counters belong to (bench.py drops a summary whose kernel id is not the loaded library's)
python - > $OUT/ids.json <<PY
import importlib, json, sys
sys.path.insert(0, "$ROOT")
L = importlib.import_module("ldpc-3gpp-matlab_amd").load()
print(json.dumps({"nrldpc_build_id": L.nrldpc_build_id().decode(), "nrldpc_kernel_id": L.nrldpc_kernel_id().decode()}))
PY
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
