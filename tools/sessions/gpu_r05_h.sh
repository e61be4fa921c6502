#!/bin/bash
# round 5, session h: refills every iteration / every second / every fourth one (DecArgs::refill_mask): parity (refill + decoder tests with each mask), then
# the parity-stop landscape per mask
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05h; mkdir -p $O; rm -rf $O/*
for m in 1 3; do
NRLDPC_REFILL_MASK=$m timeout 900 python -m pytest tests/test_refill_gpu.py -x -q -m gpu > $O/tests_mask$m.txt 2>&1; tail -2 $O/tests_mask$m.txt
done
timeout 1500 python -m pytest tests/test_refill_gpu.py tests/test_decode_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $O/tests_default.txt 2>&1; tail -2 $O/tests_default.txt
for m in 0 1 3; do
STOP=1 OUT_SUFFIX=_mask$m NRLDPC_REFILL_MASK=$m timeout 1200 python tools/bench_all_z.py > $O/stop_mask$m.log 2>&1
cp gpurun_out/bench_all_z_stop_mask$m.json $O/
done
STOP=1 OUT_SUFFIX=_policy timeout 1200 python tools/bench_all_z.py > $O/stop_policy.log 2>&1
cp gpurun_out/bench_all_z_stop_policy.json $O/
python - <<'PY'
import json, math
r={m: json.load(open("gpurun_out/r05h/bench_all_z_stop_%s.json" % m)) for m in ("mask0","mask1","mask3","policy")}
for i,x in enumerate(r["mask0"]):
    if x["Z"] <= 128: print("BG%d Z=%3d  every %.3f  2nd %.3f (%.3f)  4th %.3f (%.3f)  policy %.3f" % (x["bg"],x["Z"],x["kernel_ms"],r["mask1"][i]["kernel_ms"],r["mask1"][i]["kernel_ms"]/x["kernel_ms"],r["mask3"][i]["kernel_ms"],r["mask3"][i]["kernel_ms"]/x["kernel_ms"],r["policy"][i]["kernel_ms"]))
PY
