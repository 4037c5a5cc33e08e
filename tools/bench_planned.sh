#!/bin/bash
# The size-generic kernel families next to the tuned instantiations, one box (r04): reference tiling of a 480 x 480 crop at
# sample_point_size 60 / 96 / 160 / 480, 62 (padded to 64), and -- through the MOF_FFT_FORCE_* knobs -- the planned kernels on the
# sizes that have a tuned one (what the tuning is worth). usage (GPU box): bash tools/bench_planned.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --steps 50 --warmup 10 "$@" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["workload"][:44], "|", round(d["value"]), "pairs/s | kernel_ms", round(d["roofline"]["kernel_ms"],4))'; }
for wl in p60 p62 p96 l160 l480 ref c2; do run --workload $wl; done
echo "c2 through the planned kernel:"; MOF_FFT_FORCE_PLANNED=1 run --workload c2
echo "ref (N=120) through the planned kernel:"; MOF_FFT_FORCE_PLANNED=1 run --workload ref
echo "c2 through the large pipeline:"; MOF_FFT_FORCE_LARGE=1 run --workload c2 --batch 256
echo "ref through the large pipeline:"; MOF_FFT_FORCE_LARGE=1 run --workload ref --batch 256
