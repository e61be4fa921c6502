#!/bin/bash
# round 6, call e: (1) BG1 Z = 320 in the split form at register budgets of 4 / 5 / 6 waves per SIMD against the shipped row form (VERDICT r5
# item 7's lead), fixed 25 and the stop; (2) software prefetch in the host quantiser, paths alternated call by call; (3) the reference's
# per-step call pattern again (the refill counters now carry an event per parity-stop launch: does a small call pay for it?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06e; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
for rep in 1 2; do
  for lib in default s320w4 s320w5 s320w6; do
    if [ $lib = default ]; then timeout 120 python tools/bench_one.py 1 320 4864 0 0 25 >> $O/z320.txt 2>&1
    else NRLDPC_LIB=$PWD/exp_libs/lib_$lib.so timeout 120 python tools/bench_one.py 1 320 4864 0 0 25 >> $O/z320.txt 2>&1; fi
  done
done
for lib in "" s320w5 "" s320w5; do
  NRLDPC_LIB=${lib:+$PWD/exp_libs/lib_$lib.so} timeout 300 python tools/exp_row_refill.py 1,320 2>&1 | grep "^{" >> $O/z320.txt
done
cat $O/z320.txt
timeout 600 python tools/probe_quant_prefetch.py 2>&1 | grep "^{" | tee $O/quant_prefetch.txt | cut -c1-260
# the same with the copy threads polling for 300 us after a job before they sleep (how much of the copy / quantise phase is wake-up latency?)
NRLDPC_HOST_SPIN_US=300 timeout 600 python tools/probe_quant_prefetch.py 2>&1 | grep "^{" | sed 's/^{/{"spin_us": 300, /' | tee $O/quant_prefetch_spin300.txt | cut -c1-260
timeout 600 python tools/probe_quant_prefetch.py 2>&1 | grep "^{" | sed 's/^{/{"spin_us": 0, /' | tee -a $O/quant_prefetch.txt | cut -c1-260
g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o $O/stall_probe || exit 1
P=$O/stall_probe
run() { name=$1; shift; echo "== $name: $ARGS" >> $O/runs.txt; ( env "$@" timeout 120 $P $ARGS ) >> $O/runs.txt 2>&1; }
ARGS="f64 1 60 2 2 208 0 31 1"; run step_demo_c2_all_rows_stop NRLDPC_HOST_ZEROCOPY_KB=2048
ARGS="f64 1 60 2 2 208 -1 31 1"; run step_demo_c2_auto_stop NRLDPC_HOST_ZEROCOPY_KB=2048
ARGS="f64 1 60 1 1 384 0 27 1"; run step_r89_c1_all_rows_stop NRLDPC_HOST_ZEROCOPY_KB=2048
ARGS="f64 1 60 1 1 384 -1 27 1"; run step_r89_c1_auto_stop NRLDPC_HOST_ZEROCOPY_KB=2048
ARGS="f64 1 60 1 1 384 0 100 1"; run step_r13_c1_stop NRLDPC_HOST_ZEROCOPY_KB=2048
ARGS="f64 1 60 1 2 20 -1 22 1"; run step_cfg1_c1_auto_stop NRLDPC_HOST_ZEROCOPY_KB=2048
ARGS="f64 1 60 1 1 384 -1 27 0"; run step_r89_c1_auto_fixed25 NRLDPC_HOST_ZEROCOPY_KB=2048
grep -A1 "^==" $O/runs.txt | grep -v "^--" | paste - - | cut -c1-200
