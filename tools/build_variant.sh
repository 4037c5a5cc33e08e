#!/bin/bash
# Builds ONE variant library here (hipcc cross-compiles gfx950): the product's objects with one source recompiled under extra flags.
#   usage: tools/build_variant.sh <name> <source.hip> "<flags>"   -> tmp_ab/libmof_<name>.so   (product objects must be built: make -C mrs_optic_flow_amd/csrc)
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p $R/tmp_ab
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
EXTRA=""; [ "$SRC" == "mof_geom.hip" ] && EXTRA="-ffp-contract=off"
OTHERS=$(ls *.hip | grep -v "^$SRC$" | grep -v "^pc_kernel_quad.hip$" | sed 's/\.hip$/.o/')
hipcc --offload-arch=gfx950 $BASE $EXTRA $FLAGS -I../../include -I. -c -o $R/tmp_ab/var_$NAME.o $SRC
hipcc --offload-arch=gfx950 -shared -o $R/tmp_ab/libmof_$NAME.so $OTHERS $R/tmp_ab/var_$NAME.o -ldl
rm -f $R/tmp_ab/var_$NAME.o
echo tmp_ab/libmof_$NAME.so
