#!/bin/bash
# Same-box A/B of macro variants of ONE kernel source (boxes differ by ~5 % in clock, a box warms up over its first
# minute: only interleaved same-box numbers are comparable). Every variant is built into /tmp and selected through
# MOF_LIB_PATH; the product library and its object files are never touched (ablation variants may compute wrong results
# by design -- nothing of them can leak into a later build).
#   usage (on the GPU box):  tools/ab_variants.sh <source.hip> "<bench args>" "<flags variant 0>" "<flags variant 1>" ...
#   e.g. tools/ab_variants.sh sr_kernel.hip "--workload c5" "-DMOF_SR_FWD_ROWS=8" "-DMOF_SR_FWD_ROWS=16"
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
SRC=$1; BENCH_ARGS=$2; shift 2
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
OTHERS=$(ls *.hip | grep -v "^$SRC$" | grep -v "^pc_kernel_quad.hip$" | sed 's/\.hip$/.o/')
for o in $OTHERS; do [ -f $o ] || { echo "missing $o: build the product library first"; exit 1; }; done
EXTRA=""; [ "$SRC" == "mof_geom.hip" ] && EXTRA="-ffp-contract=off"
i=0
for V in "$@"; do
  hipcc --offload-arch=gfx950 $BASE $EXTRA $V -I../../include -I. -c -o /tmp/ab_var_$i.o $SRC
  hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_ab_$i.so $OTHERS /tmp/ab_var_$i.o -ldl
  i=$((i+1))
done
n=$i
for rep in 1 2 3; do
  for v in $(seq 0 $((n-1))); do
    line=$(MOF_LIB_PATH=/tmp/libmof_ab_$v.so python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 $BENCH_ARGS | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["roofline"]["kernel_ms"],4))')
    echo "rep $rep variant $v : $line"
  done
done
