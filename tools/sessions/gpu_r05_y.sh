#!/bin/bash
# rate matching without the division in the no-repetition path: tests, stage timing
mkdir -p gpurun_out/r05y; cd /root/repo
timeout 1500 python -m pytest tests/test_chain_gpu.py tests/test_testbench_gpu.py tests/test_harness_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05y/tests.txt
OUT_SUFFIX=_y python tools/bench_chain.py 2>&1 | grep "^{" > gpurun_out/r05y/chain.txt
python - <<'PY'
import ast
for l in open("gpurun_out/r05y/chain.txt"):
    r = ast.literal_eval(l)
    if r["stage"] in ("rate_match", "encode"): print(r["config"][:44], r["stage"][:20], round(r["ms"], 4), round(r.get("frac_of_8TBs", 0), 3))
PY
