#!/bin/bash
# One rocprofv3 PMC pass per counter group over bench.py (no trace domains combined with --pmc).
# usage: tools/pmc.sh <tag> "<group1 counters>" ["<group2 counters>" ...] -- [bench args...]
set -e
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do GROUPS_+=("$1"); shift; done
[ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${GROUPS_[@]}"; do
  rocprofv3 --pmc $g --kernel-include-regex "mof::" --output-format csv -d $OUT/pmc_g$i -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/pmc_g$i.log 2>&1 || { tail -5 $OUT/pmc_g$i.log; exit 1; }
  i=$((i+1))
done
echo "pmc groups done: $i"
