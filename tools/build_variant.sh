#!/bin/bash
# Builds ONE variant library here (hipcc cross-compiles gfx950): the product's objects with one source recompiled under extra flags.
#   usage: tools/build_variant.sh <name> <source.hip[,source2.hip ...]> "<flags>"   -> tmp_ab/libmof_<name>.so   (product objects must be built: make -C mrs_optic_flow_amd/csrc)
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p $R/tmp_ab
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
OTHERS=$(ls *.hip | grep -v "^pc_kernel_quad.hip$")
VAR=""
for S in ${SRC//,/ }; do
  EXTRA=""; [ "$S" == "mof_geom.hip" ] && EXTRA="-ffp-contract=off"
  OTHERS=$(echo "$OTHERS" | grep -v "^$S$")
  hipcc --offload-arch=gfx950 $BASE $EXTRA $FLAGS -I../../include -I. -c -o $R/tmp_ab/var_${NAME}_${S%.hip}.o $S &
  VAR="$VAR $R/tmp_ab/var_${NAME}_${S%.hip}.o"
done
wait
hipcc --offload-arch=gfx950 -shared -o $R/tmp_ab/libmof_$NAME.so $(echo "$OTHERS" | sed 's/\.hip$/.o/') $VAR -ldl
rm -f $VAR
echo tmp_ab/libmof_$NAME.so
