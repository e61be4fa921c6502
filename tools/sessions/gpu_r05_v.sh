#!/bin/bash
# channel kernel on the hardware transcendentals: tests, stage timing, Monte-Carlo loop
mkdir -p gpurun_out/r05v; cd /root/repo
timeout 1500 python -m pytest tests/test_chain_gpu.py tests/test_harness_gpu.py tests/test_system_objects_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05v/tests.txt
python tools/bench_chain.py 2>&1 | grep "^{" > gpurun_out/r05v/chain.txt
python tools/bench_montecarlo.py 2>&1 | grep "^{" > gpurun_out/r05v/mc.txt
python - <<'PY'
import ast, json
for l in open("gpurun_out/r05v/chain.txt"):
    r = ast.literal_eval(l)
    if "awgn" in r["stage"]: print(r["config"][:44], r["stage"][:20], round(r["ms"], 4), round(r.get("frac_of_8TBs", 0), 3))
for l in open("gpurun_out/r05v/mc.txt"):
    r = json.loads(l); print(r["config"], round(r["ms_median"], 3), round(r["transport_blocks_per_s"] / 1e6, 3))
PY
