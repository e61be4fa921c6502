#!/bin/bash
# Kernel experiment helper for the run-time-Z decoder: build exp_libs/lib_generic_<SUFFIX>.so from the tree's nrldpc_decode.hip
# (plus EXTRA flags) and the other objects of ldpc-3gpp-matlab_amd/build/.
set -e
R=$(cd $(dirname $0)/.. && pwd); P=$R/ldpc-3gpp-matlab_amd
mkdir -p $R/exp_libs
O=$R/exp_libs/generic_${SUFFIX:-x}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -enable-post-misched=false -I$R/include -I$P/csrc $EXTRA -c ${SRC:-$P/csrc/nrldpc_decode.hip} -o $O
OBJS=$(ls $P/build/*.o | grep -v "/nrldpc_decode.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O -o $R/exp_libs/lib_generic_${SUFFIX:-x}.so
rm -f $O
echo built lib_generic_${SUFFIX:-x}.so
