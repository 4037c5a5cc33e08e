#!/bin/bash
# Per-kernel average durations of one bench workload (rocprofv3 --kernel-trace --stats): tools/stats_workload.sh <workload> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stw && rocprofv3 --kernel-trace --stats --kernel-include-regex "mof::" --output-format csv -d /tmp/stw -- python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload $WL --steps 10 --warmup 3 "$@" > /tmp/stw.log 2>&1
python3 - $(find /tmp/stw -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mof::" in r["Name"]: print(f'{float(r["AverageNs"])/1e3:10.1f} us x {int(r["Calls"]):5d}  {float(r["Percentage"]):5.1f} %  {r["Name"][:120]}')
PY
