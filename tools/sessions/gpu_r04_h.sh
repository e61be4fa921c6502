#!/bin/bash
# GPU session: interleaved block geometry -- parity and A/B against the kernels that served the same sizes before, at a fixed SNR
# (all rows) and at each (size, layer count)'s own waterfall.  FILTER: python expression over (bg, z, n, m) selecting list entries
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/ab_ilv6_*.jsonl gpurun_out/ab_ilv7_*.jsonl
export FILTER=${FILTER:-True}
ALL=$(python - <<'PY'
import importlib, os
b = importlib.import_module("ldpc-3gpp-matlab_amd.build")
E = [(bg, z) for bg, z, n, m in b.Z64I if eval(os.environ["FILTER"])]
print(" ".join("%d,%d" % e for e in E))
print(" ".join("%d,%d,%d" % (bg, z, 13 if bg == 1 else 17) for bg, z in E) + " " + " ".join("%d,%d,%d" % (bg, z, 24 if bg == 1 else 7) for bg, z in E))
PY
)
A=$(echo "$ALL" | sed -n 1p); N1=$(echo "$ALL" | sed -n 2p)
timeout 900 python tools/ab_ilv.py gpurun_out/ab_ilv6_new.jsonl $A > gpurun_out/ab_ilv6_new.log 2>&1
grep -c FAIL gpurun_out/ab_ilv6_new.log
NO_CHECK=1 NRLDPC_NO_ILV=1 timeout 900 python tools/ab_ilv.py gpurun_out/ab_ilv6_old.jsonl $A > gpurun_out/ab_ilv6_old.log 2>&1
WATERFALL=1 timeout 1500 python tools/ab_ilv.py gpurun_out/ab_ilv7_new.jsonl $A $N1 > gpurun_out/ab_ilv7_new.log 2>&1
grep -c FAIL gpurun_out/ab_ilv7_new.log
WATERFALL=1 NO_CHECK=1 NRLDPC_NO_ILV=1 timeout 1500 python tools/ab_ilv.py gpurun_out/ab_ilv7_old.jsonl $A $N1 > gpurun_out/ab_ilv7_old.log 2>&1
tail -1 gpurun_out/ab_ilv7_old.log
