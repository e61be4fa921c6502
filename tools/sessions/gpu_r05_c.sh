#!/bin/bash
# round 5, session c: the rebuilt library (ADVICE fixes, per-unit name spaces, ABI revision 5, copy-out overlapped with quantisation):
# decoder tests + layers + abi caller + full-size tests; host-path timings incl. R = 8/9 doubles with and without AUTO
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05c; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o $O/stall_probe || exit 1
P=$O/stall_probe
run() { name=$1; shift; echo "== $name: $ARGS" >> $O/runs.txt; ( NRLDPC_HOST_TRACE=1 timeout 120 $P $ARGS ) >> $O/runs.txt 2>&1; }
ARGS="f16 0 10"; run f16_bytes
ARGS="f64 0 10"; run f64_bytes
ARGS="f16 1 10"; run f16_packed
ARGS="f64 1 10"; run f64_packed
ARGS="f32 1 10"; run f32_packed
ARGS="f64 1 10 4096 1 384 0 27"; run f64_packed_r89_all_rows
ARGS="f64 1 10 4096 1 384 -1 27"; run f64_packed_r89_auto
ARGS="f64 1 10 4096 1 384 5 27"; run f64_packed_r89_explicit5
ARGS="f32 1 10 4096 1 384 -1 27"; run f32_packed_r89_auto
ARGS="f64 1 10 8192 2 208 0 31"; run f64_packed_demo_bg2_z208_all_rows
ARGS="f64 1 10 8192 2 208 -1 31"; run f64_packed_demo_bg2_z208_auto
grep -h "^==\|^call  [5-9]\|layers of" $O/runs.txt
timeout 2400 python -m pytest tests/test_layers_gpu.py tests/test_abi_caller_gpu.py tests/test_decode_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -12 $O/tests.txt
