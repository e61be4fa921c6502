#!/bin/bash
# round-4 GPU session B: decoder tests (CRC stop, packed output), CRC-stop measurement, run-time layer counts on the packed sizes,
# ET on the new BG1 packed sizes, host path with / without the chunk ramp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r04b_pytest.log
python tools/bench_crc_stop.py > gpurun_out/r04b_crc_stop.log 2>&1; tail -5 gpurun_out/r04b_crc_stop.log | cut -c1-600
export NL_ZS=8,32,56,80,96,352 NL_NLS=5,13,24,0
NRLDPC_NO_PRUNED_PIPELINE=1 python tools/bench_nl.py rt > gpurun_out/r04b_nl_rt.log 2>&1; mv gpurun_out/bench_nl_rt.json gpurun_out/bench_nl_packed_rt.json
NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_NO_RT=1 python tools/bench_nl.py general > gpurun_out/r04b_nl_general.log 2>&1; mv gpurun_out/bench_nl_general.json gpurun_out/bench_nl_packed_general.json
for z in 88 96 176 352; do
  python tools/exp_check.py 1 $z 2>&1 | grep -E "Gbit|FAIL" | sed "s/^/packed   /" | tee -a gpurun_out/r04b_et_packed.log
  NRLDPC_NO_PACKED=1 python tools/exp_check.py 1 $z 2>&1 | grep -E "Gbit|FAIL" | sed "s/^/previous /" | tee -a gpurun_out/r04b_et_packed.log
done
NRLDPC_HOST_TRACE=1 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-early-term > gpurun_out/r04b_bench_ramp.json 2> gpurun_out/r04b_bench_ramp.err
NRLDPC_HOST_RAMP=0 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-early-term > gpurun_out/r04b_bench_noramp.json 2>/dev/null
python - <<'PY'
import json
for f in ("ramp","noramp"):
    d=json.load(open("gpurun_out/r04b_bench_%s.json"%f))
    print(f, {k:(round(v["ms_median"],3),round(v["ms_min"],3),round(v["ms_max"],3),round(v["value"],2)) for k,v in d["e2e"].items() if isinstance(v,dict)})
PY
