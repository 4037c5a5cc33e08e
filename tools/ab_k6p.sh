#!/bin/bash
# K6p (sr_cols_split_kernel: two columns per wave, radix-32 stage on lane pairs) against K6s, same box: c5 / c5seq rates
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for wl in c5 c5seq; do
    for v in 0 1; do
      line=$(MOF_SR_COLS_SPLIT=$v python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload $wl --steps 20 --warmup 5 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))')
      echo "rep $rep $wl split=$v : $line"
    done
  done
done
