#!/bin/bash
# Same-box A/B of ENVIRONMENT knobs of one library (run-time-selected kernel forms): alternates the settings three times.
#   usage (on the GPU box): tools/ab_env.sh "<bench args>" "<env 0>" "<env 1>" ...     e.g. tools/ab_env.sh "--workload c5" "MOF_X=0" "MOF_SR_LP_B64=1"
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
for rep in 1 2 3; do
  i=0
  for E in "$@"; do
    line=$(env $E python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 $ARGS 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["roofline"]["kernel_ms"],4))')
    echo "rep $rep [$E] : $line"
    i=$((i+1))
  done
done
