#!/bin/bash
# round 5, session g: claim-ahead refill + zero-copy small calls: the whole GPU suite, the reference's per-step call pattern with and
# without zero-copy, the parity-stop landscape again (against session final-1's, normalised by the sizes that do not refill)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05g; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > $O/gputests.log 2>&1; cat $O/gputests.log
g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o $O/stall_probe || exit 1
P=$O/stall_probe
run() { name=$1; shift; echo "== $name: $ARGS" >> $O/runs.txt; ( env "$@" timeout 120 $P $ARGS ) >> $O/runs.txt 2>&1; }
for zc in 2048 0; do
ARGS="f64 1 60 2 2 208 0 31"; run step_demo_c2_all_rows_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 2 2 208 -1 31"; run step_demo_c2_auto_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 1 384 0 27"; run step_r89_c1_all_rows_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 1 384 -1 27"; run step_r89_c1_auto_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 0 60 1 1 384 0"; run step_r13_c1_bytes_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 8 1 384 0"; run step_r13_c8_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
ARGS="f64 1 60 1 2 20 -1 22"; run step_cfg1_c1_auto_zc$zc NRLDPC_HOST_ZEROCOPY_KB=$zc
done
grep -h "^==\|^call 5[7-9]" $O/runs.txt
STOP=1 OUT_SUFFIX=_g timeout 1500 python tools/bench_all_z.py > $O/stop.log 2>&1
cp gpurun_out/bench_all_z_stop_g.json $O/
tail -3 $O/stop.log
