#!/bin/bash
# round 5, session d: slot refill under the parity stop (nrldpc_decode_z64p.h) -- parity, then its effect on the sizes whose parity
# stop the packed / interleaved kernels already serve (STOP=1 landscape with and without NRLDPC_NO_REFILL); AUTO scan timing again
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05d; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 1500 python -m pytest tests/test_refill_gpu.py -x -q -m gpu --durations=5 > $O/tests_refill.txt 2>&1; tail -12 $O/tests_refill.txt
timeout 1500 python -m pytest tests/test_decode_gpu.py -x -q -m gpu -k "interleaved or every_lifting or iteration_by_iteration" --durations=5 > $O/tests_decode.txt 2>&1; tail -5 $O/tests_decode.txt
g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o $O/stall_probe || exit 1
P=$O/stall_probe
run() { name=$1; shift; echo "== $name: $ARGS" >> $O/runs.txt; ( NRLDPC_HOST_TRACE=1 timeout 120 $P $ARGS ) >> $O/runs.txt 2>&1; }
ARGS="f64 1 10"; run f64_packed
ARGS="f64 1 10 4096 1 384 0 27"; run f64_packed_r89_all_rows
ARGS="f64 1 10 4096 1 384 -1 27"; run f64_packed_r89_auto
ARGS="f64 1 10 4096 1 384 5 27"; run f64_packed_r89_explicit5
ARGS="f32 1 10 4096 1 384 -1 27"; run f32_packed_r89_auto
ARGS="f64 1 10 8192 2 208 0 31"; run f64_packed_demo_bg2_z208_all_rows
ARGS="f64 1 10 8192 2 208 -1 31"; run f64_packed_demo_bg2_z208_auto
grep -h "^==\|^call  [7-9]\|layers of" $O/runs.txt
STOP=1 OUT_SUFFIX=_refill timeout 1200 python tools/bench_all_z.py > $O/stop_refill.log 2>&1
STOP=1 OUT_SUFFIX=_norefill NRLDPC_NO_REFILL=1 timeout 1200 python tools/bench_all_z.py > $O/stop_norefill.log 2>&1
cp gpurun_out/bench_all_z_stop_refill.json gpurun_out/bench_all_z_stop_norefill.json $O/
python - <<'PY'
import json
a=json.load(open("gpurun_out/r05d/bench_all_z_stop_refill.json")); b=json.load(open("gpurun_out/r05d/bench_all_z_stop_norefill.json"))
for x,y in zip(a,b):
    print("BG%d Z=%3d refill %.3f ms  no refill %.3f ms  ratio %.3f  iters %.2f" % (x["bg"],x["Z"],x["kernel_ms"],y["kernel_ms"],x["kernel_ms"]/y["kernel_ms"],x["mean_iters"]))
PY
