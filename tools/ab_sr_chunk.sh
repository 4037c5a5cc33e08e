#!/bin/bash
# Same-box sweep of the estimator's pass size (MOF_SR_CHUNK pairs per pass) with and without non-temporal stream accesses:
# does an intermediate that fits the 256 MB Infinity Cache (64 pairs: Zt 118 MB + Dt 59 MB) beat the 512-pair default?
#   usage (GPU box): tools/ab_sr_chunk.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
for f in sr_kernel sr_seq_kernel; do hipcc --offload-arch=gfx950 $BASE -DMOF_SR_NT=0 -I../../include -I. -c -o /tmp/nt0_$f.o $f.hip; done
hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_nt0.so $(ls *.o | grep -v "^sr_kernel.o$" | grep -v "^sr_seq_kernel.o$") /tmp/nt0_sr_kernel.o /tmp/nt0_sr_seq_kernel.o
for wl in ${WLS:-c5 c5seq}; do
  for v in nt plain; do
    for chunk in ${CHUNKS:-32 64 128 256 512}; do
      LIB=$R/mrs_optic_flow_amd/libmof_hip.so; [ $v == plain ] && LIB=/tmp/libmof_nt0.so
      echo "$wl $v chunk $chunk $(MOF_SR_CHUNK=$chunk MOF_LIB_PATH=$LIB python3 $R/bench.py --workload $wl --no-cpu-baseline --no-others --sustain-s 0 --steps 60 --warmup 15 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))')"
    done
  done
done
