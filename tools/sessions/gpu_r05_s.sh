#!/bin/bash
# rate recovery with every load of a thread issued before its first store: tests, stage timings
mkdir -p gpurun_out/r05s; cd /root/repo
timeout 1200 python -m pytest tests/test_chain_gpu.py tests/test_testbench_gpu.py tests/test_harness_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05s/tests.txt
python tools/bench_chain.py 2>&1 | grep "^{" > gpurun_out/r05s/chain.txt
NRLDPC_RR_GENERAL=1 OUT_SUFFIX=_general python tools/bench_chain.py 2>&1 | grep "^{" > gpurun_out/r05s/chain_general.txt
python - <<'PY'
import ast
for f in ("chain", "chain_general"):
    print("==", f)
    for l in open("gpurun_out/r05s/%s.txt" % f):
        r = ast.literal_eval(l)
        if "rate_recover" in r["stage"] or "receive chain" in r["stage"]: print(r["config"][:40], r["stage"][:30], round(r["ms"], 4), round(r.get("frac_of_8TBs", 0), 3))
PY
