#!/bin/bash
# Round-3 GPU session: full GPU suite on the shipped library, A/B probes on the two-form library, bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/gputests.log 2>&1; cat gpurun_out/gputests.log
AB=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip_ab.so
if [ -f $AB ]; then
  for nl in 13 24 5; do for s in 0 1; do
    NRLDPC_LIB=$AB NRLDPC_SPLIT=$s ESN0=3.0 python tools/exp_check.py 1 384 $nl 2>&1 | grep -E "Gbit|FAIL|rror" | sed "s/^/split=$s /" | tee -a gpurun_out/forms_nl.log
  done; done
  # per-iteration cost of the early-termination build when nothing converges (-3 dB): ET vs fixed, both forms
  for s in 0 1; do NRLDPC_LIB=$AB NRLDPC_SPLIT=$s ESN0=-3.0 python tools/exp_check.py 1 384 2>&1 | grep -E "Gbit|FAIL|rror" | sed "s/^/split=$s noconv /" | tee -a gpurun_out/forms_nl.log; done
fi
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_line.json; cut -c1-600 gpurun_out/bench_line.json
