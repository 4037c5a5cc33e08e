#!/bin/bash
# Same-box A/B of PREBUILT variant libraries (built here with hipcc into tmp_ab/, which travels to the GPU box with the snapshot): alternates
# them three times over one bench command. Building here instead of on the box saves GPU-minutes (tools/ab_variants.sh builds on the box).
#   usage (GPU box): tools/ab_libs.sh "<bench args>" <lib 0> <lib 1> ...      [ENV: any MOF_* knob, e.g. MOF_FFT_HALF=1]
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
for rep in 1 2 3; do
  for L in "$@"; do
    line=$(MOF_LIB_PATH=$R/$L python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 $ARGS 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["roofline"]["kernel_ms"],4))')
    echo "rep $rep [$L] : $line"
  done
done
