#!/bin/bash
# c5 / c5seq with K56 (sr_fused_kernel.hip: row transforms on the matrix cores inside the column kernel, no Zh in HBM) against K5s + K6s:
# same-box interleaved rates
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for wl in ${AB_WORKLOADS:-c5}; do
    for v in 0 1; do
      line=$(MOF_SR_FUSED=$v python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload $wl --steps 20 --warmup 5 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))')
      echo "rep $rep $wl fused=$v : $line"
    done
  done
done
