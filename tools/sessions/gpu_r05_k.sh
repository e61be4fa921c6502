#!/bin/bash
# round 5, session k: small host calls pack their bits on the CPU (one launch per call): tests that go through the small host path, and the
# reference's per-step pattern with the parity stop (the gateway's mode) -- every row / AUTO, zero-copy on / off
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05k; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 ) > $O/gputests.log 2>&1; cat $O/gputests.log
g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o $O/stall_probe || exit 1
P=$O/stall_probe
run() { name=$1; shift; echo "== $name: $ARGS" >> $O/runs.txt; ( env "$@" timeout 120 $P $ARGS ) >> $O/runs.txt 2>&1; }
for zc in 2048 0; do
ARGS="f64 1 60 2 2 208 0 31 1"; run step_demo_c2_all_rows_stop_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 2 2 208 -1 31 1"; run step_demo_c2_auto_stop_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 1 384 0 27 1"; run step_r89_c1_all_rows_stop_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 1 384 -1 27 1"; run step_r89_c1_auto_stop_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 1 384 0 100 1"; run step_r13_c1_stop_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 2 20 -1 22 1"; run step_cfg1_c1_auto_stop_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 1 384 -1 27 0"; run step_r89_c1_auto_fixed25_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
done
grep -h "^==\|^call 59" $O/runs.txt | paste - - | awk '{print $2, "->", $(NF-3), $(NF-2)}'
