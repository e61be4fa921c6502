#!/bin/bash
# Kernel experiment helper: build a library variant whose (BG, Z[, NL]) compile-time decoder uses a different
# codewords-per-workgroup / waves-per-SIMD setting or extra -D flags.
# usage: [EXTRA="-D..."] [SUFFIX=name] tools/exp_z64.sh <bg> <z> <ncwg> <wpe> [nl]  -> exp_libs/lib_<bg>_<z>[_nl<nl>]_<ncwg>_<wpe>[_<SUFFIX>].so
# PACKED=1: the packed-geometry unit of that (BG, Z) instead (nrldpc_decode_z64p_inst.hip; <ncwg> <wpe> are only part of the name)
# (the other objects come from ldpc-3gpp-matlab_amd/build/: build the default library first)
set -e
BG=$1; Z=$2; N=$3; W=$4; NL=$5
R=$(cd $(dirname $0)/.. && pwd); P=$R/ldpc-3gpp-matlab_amd
mkdir -p $R/exp_libs
TAG=${BG}_${Z}${NL:+_nl$NL}
NAME=${TAG}_${N}_${W}${SUFFIX:+_$SUFFIX}
O=$R/exp_libs/z64_${NAME}.o
if [ -n "$PACKED" ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -enable-post-misched=false -I$R/include -I$P/csrc -DNRLDPC_Z64_BG=$BG -DNRLDPC_Z64_Z=$Z $EXTRA -c $P/csrc/nrldpc_decode_z64p_inst.hip -o $O
OBJS=$(ls $P/build/*.o | grep -v "z64p_${TAG}.o")
else
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -enable-post-misched=false -I$R/include -I$P/csrc -DNRLDPC_Z64_BG=$BG -DNRLDPC_Z64_Z=$Z ${NL:+-DNRLDPC_Z64_NL=$NL} -DNRLDPC_Z64_NCWG=$N -DNRLDPC_Z64_WPE=$W $EXTRA -c $P/csrc/nrldpc_decode_z64_inst.hip -o $O
OBJS=$(ls $P/build/*.o | grep -v "z64_${TAG}.o")
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O -o $R/exp_libs/lib_${NAME}.so
rm -f $O
echo built lib_${NAME}.so
