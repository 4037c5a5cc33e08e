#!/bin/bash
# Per-kernel average durations of the variants tools/ab_variants.sh left in /tmp (libmof_ab_<i>.so): one rocprofv3 --kernel-trace --stats
# run per variant over the same bench command.   usage (GPU box, after ab_variants.sh): tools/ab_stats.sh "<bench args>" <n variants>
R=${GRAFT_REPO_ROOT:-/root/repo}
BENCH_ARGS=$1; N=${2:-2}
cd /tmp && export TMPDIR=/tmp
for v in $(seq 0 $((N-1))); do
  rm -rf /tmp/abst_$v
  MOF_LIB_PATH=/tmp/libmof_ab_$v.so rocprofv3 --kernel-trace --stats --kernel-include-regex "mof::" --output-format csv -d /tmp/abst_$v -- python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 $BENCH_ARGS > /tmp/abst_$v.log 2>&1
  echo "== variant $v"
  python3 - $(find /tmp/abst_$v -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{float(r["AverageNs"])/1e3:10.1f} us x {int(r["Calls"]):5d}  {float(r["Percentage"]):5.1f} %  {r["Name"][:110]}')
PY
done
