#!/bin/bash
# Same-box A/B for c4 (128 x 128 patches): the packed pair kernel (one workgroup per CU) against the pair kernel on the HALF tile
# (pc_seq_half.hip: pc_pair_half_kernel, MOF_FFT_PAIR_HALF=1, two workgroups per CU). usage (on the GPU box): bash tools/ab_pair_half.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  line=$(env $2 python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --steps 30 --warmup 10 --workload c4 --batch 128 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["roofline"]["kernel_ms"],4))')
  echo "$1: $line"
}
for rep in 1 2 3; do
  run "packed (default)" MOF_X=1
  run "pair-half 2 wg/cu" MOF_FFT_PAIR_HALF=1
  run "pair-half 1 wg/cu" "MOF_FFT_PAIR_HALF=1 MOF_FFT_PAIR_HALF_WGS=1"
done
