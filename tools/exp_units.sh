#!/bin/bash
# Kernel experiment helper (round 6): ONE library in which several (BG, Z) units of the block-geometry decoder are rebuilt with extra -D flags
# (tools/exp_z64.sh replaces a single unit).  usage: EXTRA="-D..." tools/exp_units.sh <name> <bg>,<z> [<bg>,<z> ...]  -> exp_libs/lib_<name>.so
# (the other objects come from ldpc-3gpp-matlab_amd/build/: build the default library first)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd); P=$R/ldpc-3gpp-matlab_amd
mkdir -p $R/exp_libs
OBJS=$(ls $P/build/*.o)
NEW=""
for u in "$@"; do
  BG=${u%,*}; Z=${u#*,}
  O=$R/exp_libs/z64_${NAME}_${BG}_${Z}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -enable-post-misched=false -I$R/include -I$P/csrc -DNRLDPC_Z64_BG=$BG -DNRLDPC_Z64_Z=$Z -DNRLDPC_UNIT=u_z64_${BG}_${Z} $EXTRA -c $P/csrc/nrldpc_decode_z64_inst.hip -o $O &
  OBJS=$(echo "$OBJS" | grep -v "/z64_${BG}_${Z}.o")
  NEW="$NEW $O"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $NEW -o $R/exp_libs/lib_${NAME}.so
rm -f $NEW
echo built lib_${NAME}.so
