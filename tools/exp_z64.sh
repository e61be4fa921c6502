#!/bin/bash
# Kernel experiment helper: build a library variant whose (BG, Z) compile-time decoder uses a different
# codewords-per-workgroup / waves-per-SIMD setting.  usage: tools/exp_z64.sh <bg> <z> <ncwg> <wpe> -> exp_libs/lib_<bg>_<z>_<ncwg>_<wpe>.so
set -e
BG=$1; Z=$2; N=$3; W=$4
R=$(cd $(dirname $0)/.. && pwd); P=$R/ldpc-3gpp-matlab_amd
O=$R/exp_libs/z64_${BG}_${Z}_${N}_${W}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$P/csrc -DNRLDPC_Z64_BG=$BG -DNRLDPC_Z64_Z=$Z -DNRLDPC_Z64_NCWG=$N -DNRLDPC_Z64_WPE=$W $EXTRA -c $P/csrc/nrldpc_decode_z64_inst.hip -o $O
OBJS=$(ls $P/build/*.o | grep -v "z64_${BG}_${Z}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O -o $R/exp_libs/lib_${BG}_${Z}_${N}_${W}.so
echo built lib_${BG}_${Z}_${N}_${W}.so
