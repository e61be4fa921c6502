#!/bin/bash
# lead for the next round: EVERY interleaved entry with the parity stop (an out-of-tree build with all modes, exp_libs/lib_allmodes.so)
# against the previous kernels, below / at / above each size's waterfall
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/ab_wf_*.jsonl
A=$(python - <<'PY'
import importlib
b = importlib.import_module("ldpc-3gpp-matlab_amd.build")
print(" ".join("%d,%d" % (bg, z) for bg, z, _, _ in b.Z64I))
PY
)
for off in -2.0 0 1.5 4.0; do
WATERFALL=1 WF_OFFSET=$off NO_CHECK=1 NRLDPC_LIB=$PWD/exp_libs/lib_allmodes.so timeout 600 python tools/ab_ilv.py gpurun_out/ab_wf_new_$off.jsonl $A > gpurun_out/ab_wf_new_$off.log 2>&1
WATERFALL=1 WF_OFFSET=$off NO_CHECK=1 NRLDPC_NO_ILV=1 timeout 600 python tools/ab_ilv.py gpurun_out/ab_wf_old_$off.jsonl $A > gpurun_out/ab_wf_old_$off.log 2>&1
done
tail -1 gpurun_out/ab_wf_old_4.0.log
