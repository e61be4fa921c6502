#!/bin/bash
# round 6, session p: layer-count classes (run-time prefix over the tables of the first c rows) for BG1 / BG2 Z = 384 against the listed
# compile-time builds and the all-rows run-time-prefix kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06p; mkdir -p $O
L=$PWD/exp_libs; D=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip.so
export NL_ZS=384
NL_NLS=5,7,9,12,13,17,22,24,32 NRLDPC_LIB=$D python tools/bench_nl.py default > $O/default.log 2>&1
NL_NLS=5,7,9,12,13,17,22,24,32 NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_LIB=$D python tools/bench_nl.py rt > $O/rt.log 2>&1
for spec in "32:32,24,22,17" "24:24,22,17,13" "17:17,13,12" "16:13,12,9" "12:12,9,7" "8:7,5"; do
  c=${spec%%:*}; nls=${spec#*:}
  [ $c = 32 ] && units="2,384"; [ $c = 24 ] && units="1,384 2,384"; [ $c = 17 ] && units="2,384"; [ $c = 16 ] && units="1,384 2,384"; [ $c = 12 ] && units="2,384"; [ $c = 8 ] && units="1,384 2,384"
  NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_LIB=$L/lib_rtc$c.so timeout 600 python tools/exp_classes.py $c $units > $O/parity_rtc$c.log 2>&1
  NL_NLS=$nls NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_LIB=$L/lib_rtc$c.so timeout 600 python tools/bench_nl.py rtc$c > $O/rtc$c.log 2>&1
done
cp gpurun_out/bench_nl_*.json $O/
grep -h "FAIL\|Error\|error" $O/*.log | head -20
grep -c "parity ok" $O/parity_*.log
