#!/bin/bash
# c5 / c5seq with the FFT field (K1, VALU/LDS-bound) on a side stream under the estimator's HBM-bound kernels: same-box interleaved A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for wl in c5 c5seq; do
    for v in 0 1; do
      line=$(MOF_BENCH_FFT_SIDE_STREAM=$v python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload $wl --steps 20 --warmup 5 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))')
      echo "rep $rep $wl side_stream=$v : $line"
    done
  done
done
