#!/bin/bash
# Same-box A/B of the fused half-tile kernel (csrc/pc_half_kernel.hip, MOF_FFT_HALF) against the paths it replaces / competes with:
#   l160 (480^2, 3 x 3 of 160^2): MOF_FFT_HALF=0 = the four-kernel pipeline through HBM scratch (r04) | default = half-tile kernel
#   ref (N = 120) and c4 (N = 128): default = the tuned pair kernels, one workgroup per CU | MOF_FFT_HALF=1 = half-tile kernel, two per CU
# usage (on the GPU box): bash tools/ab_half.sh [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-50}
run() { # label env workload extra
  line=$(env $2 python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --steps $STEPS --warmup 10 --workload $3 $4 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["roofline"]["kernel_ms"],4))')
  echo "$1 $3: $line"
}
for rep in 1 2 3; do
  run "pipeline(HALF=0)" MOF_FFT_HALF=0 l160
  run "half-tile(default)" MOF_X=1 l160
  run "tuned(default)" MOF_X=1 ref
  run "half-tile(HALF=1)" MOF_FFT_HALF=1 ref
  run "tuned(default)" MOF_X=1 c4 "--batch 128"
  run "half-tile(HALF=1)" MOF_FFT_HALF=1 c4 "--batch 128"
  run "tuned(default)" MOF_X=1 c2
  run "half-tile(HALF=1)" MOF_FFT_HALF=1 c2
done
