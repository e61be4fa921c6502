#!/bin/bash
# round 5, session e: (1) every interleaved entry with the parity stop + slot refill (all-modes build) against the kernels that serve
# the size now, at each size's waterfall, all rows and a pruned count -> the mode bits of NRLDPC_Z64I_LIST; (2) headline A/B: dual
# rows extended beyond rows 0-3 (timing-only builds); (3) AUTO scan timing; the reference's own call pattern (C code blocks per step)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05e; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
# (2) headline: alternate twice
for rep in 1 2; do
  for lib in default exp_libs/lib_dual5.so exp_libs/lib_dual13.so; do
    if [ $lib = default ]; then timeout 120 python tools/bench_one.py 1 384 4096 0 0 25 >> $O/headline_dual.txt 2>&1
    else NRLDPC_LIB=$PWD/$lib timeout 120 python tools/bench_one.py 1 384 4096 0 0 25 >> $O/headline_dual.txt 2>&1; fi
  done
done
cat $O/headline_dual.txt
# (3)
g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o $O/stall_probe || exit 1
P=$O/stall_probe
run() { name=$1; shift; echo "== $name: $ARGS" >> $O/runs.txt; ( NRLDPC_HOST_TRACE=1 timeout 120 $P $ARGS ) >> $O/runs.txt 2>&1; }
ARGS="f64 1 10 4096 1 384 0 27"; run f64_packed_r89_all_rows
ARGS="f64 1 10 4096 1 384 -1 27"; run f64_packed_r89_auto
ARGS="f64 1 10 4096 1 384 5 27"; run f64_packed_r89_explicit5
ARGS="f32 1 10 4096 1 384 0 27"; run f32_packed_r89_all_rows
ARGS="f32 1 10 4096 1 384 -1 27"; run f32_packed_r89_auto
ARGS="f16 1 10 4096 1 384 0 27"; run f16_packed_r89_all_rows
ARGS="f16 1 10 4096 1 384 -1 27"; run f16_packed_r89_auto
# the reference's own pattern: C code blocks per step() (plot_BLER_vs_SNR.m defaults: BG2 Z=208, C = 2, 21 of 42 rows; and one BG1 Z=384 R=8/9 block)
ARGS="f64 1 40 2 2 208 0 31"; run step_demo_c2_all_rows
ARGS="f64 1 40 2 2 208 -1 31"; run step_demo_c2_auto
ARGS="f64 1 40 1 1 384 0 27"; run step_r89_c1_all_rows
ARGS="f64 1 40 1 1 384 -1 27"; run step_r89_c1_auto
grep -h "^==\|^call  [7-9] \|^call 3[7-9]\|layers of" $O/runs.txt
# (1)
A=$(python - <<'PY'
import importlib
b = importlib.import_module("ldpc-3gpp-matlab_amd.build")
print(" ".join("%d,%d" % (bg, z) for bg, z, _, _ in b.Z64I))
PY
)
AP=$(python - <<'PY'
import importlib
b = importlib.import_module("ldpc-3gpp-matlab_amd.build")
print(" ".join("%d,%d,%d" % (bg, z, 13 if bg == 1 else 9) for bg, z, _, _ in b.Z64I))
PY
)
WATERFALL=1 NO_CHECK=1 NRLDPC_LIB=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip_allmodes.so timeout 900 python tools/ab_ilv.py $O/ab_new_all.jsonl $A > $O/ab_new_all.log 2>&1
WATERFALL=1 NO_CHECK=1 NRLDPC_NO_ILV=1 timeout 900 python tools/ab_ilv.py $O/ab_old_all.jsonl $A > $O/ab_old_all.log 2>&1
WATERFALL=1 NO_CHECK=1 NRLDPC_LIB=$PWD/ldpc-3gpp-matlab_amd/libnrldpc_hip_allmodes.so timeout 900 python tools/ab_ilv.py $O/ab_new_pr.jsonl $AP > $O/ab_new_pr.log 2>&1
WATERFALL=1 NO_CHECK=1 NRLDPC_NO_ILV=1 timeout 900 python tools/ab_ilv.py $O/ab_old_pr.jsonl $AP > $O/ab_old_pr.log 2>&1
wc -l $O/*.jsonl
