#!/bin/bash
# Same-box A/B of the estimator kernels with and without non-temporal stream accesses (MOF_SR_NT, sr_common.hpp).
#   usage (GPU box): tools/ab_sr_nt.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
for f in sr_kernel sr_seq_kernel; do hipcc --offload-arch=gfx950 $BASE -DMOF_SR_NT=0 -I../../include -I. -c -o /tmp/nt0_$f.o $f.hip; done
hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_nt0.so $(ls *.o | grep -v "^sr_kernel.o$" | grep -v "^sr_seq_kernel.o$") /tmp/nt0_sr_kernel.o /tmp/nt0_sr_seq_kernel.o
for wl in c5 c5seq; do
  for rep in 1 2 3; do
    for v in plain nt; do
      LIB=$R/mrs_optic_flow_amd/libmof_hip.so; [ $v == plain ] && LIB=/tmp/libmof_nt0.so
      echo "$wl $v $(MOF_LIB_PATH=$LIB python3 $R/bench.py --workload $wl --no-cpu-baseline --no-others --sustain-s 0 --steps 100 --warmup 20 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))')"
    done
  done
done
