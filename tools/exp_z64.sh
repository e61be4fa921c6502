#!/bin/bash
# Kernel experiment helper: build a library variant whose (BG, Z[, NL]) compile-time decoder uses a different
# codewords-per-workgroup / waves-per-SIMD setting.
# usage: tools/exp_z64.sh <bg> <z> <ncwg> <wpe> [nl]  -> exp_libs/lib_<bg>_<z>[_nl<nl>]_<ncwg>_<wpe>.so    (EXTRA="-D..." adds flags)
set -e
BG=$1; Z=$2; N=$3; W=$4; NL=$5
R=$(cd $(dirname $0)/.. && pwd); P=$R/ldpc-3gpp-matlab_amd
mkdir -p $R/exp_libs
TAG=${BG}_${Z}${NL:+_nl$NL}
O=$R/exp_libs/z64_${TAG}_${N}_${W}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$P/csrc -DNRLDPC_Z64_BG=$BG -DNRLDPC_Z64_Z=$Z ${NL:+-DNRLDPC_Z64_NL=$NL} -DNRLDPC_Z64_NCWG=$N -DNRLDPC_Z64_WPE=$W $EXTRA -c $P/csrc/nrldpc_decode_z64_inst.hip -o $O
OBJS=$(ls $P/build/*.o | grep -v "z64_${TAG}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O -o $R/exp_libs/lib_${TAG}_${N}_${W}.so
if [ -n "$SUFFIX" ]; then mv $R/exp_libs/lib_${TAG}_${N}_${W}.so $R/exp_libs/lib_${TAG}_${N}_${W}_$SUFFIX.so; fi
echo built lib_${TAG}_${N}_${W}${SUFFIX:+_$SUFFIX}.so
