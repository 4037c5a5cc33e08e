#!/bin/bash
# Same-box sweep of the c5 pipeline's diagnostic knobs: frame pairs per pipeline pass (MOF_SR_CHUNK), images per remap
# wave (MOF_SR_LP_IPW), the two-lane overlap of remaps and transforms (MOF_SR_OVERLAP).
# usage (on the GPU box): bash tools/sweep_c5.sh > gpurun_out/sweep_c5.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
for ov in 0 1; do
  for chunk in 64 128 256; do
    for ipw in 16 32; do
      v=$(MOF_SR_OVERLAP=$ov MOF_SR_CHUNK=$chunk MOF_SR_LP_IPW=$ipw python3 $R/bench.py --workload c5 --no-cpu-baseline --sustain-s 0 --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))")
      echo "overlap $ov chunk $chunk ipw $ipw : $v"
    done
  done
done
