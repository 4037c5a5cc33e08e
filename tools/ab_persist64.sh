#!/bin/bash
# N = 64 in the persistent + prefetch form (MOF_PERSIST_MIN_N=64) with the co-resident workgroups' phases staggered
# (MOF_PC_STAGGER units of 6400 clocks), against the product's one-workgroup-per-patch form. usage (GPU box): tools/ab_persist64.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
hipcc --offload-arch=gfx950 $BASE -DMOF_PERSIST_MIN_N=64 -I../../include -I. -c -o /tmp/p64.o pc_kernel.hip
hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_p64.so $(ls *.o | grep -v "^pc_kernel.o$") /tmp/p64.o
run() { python3 $R/bench.py --workload c2 --no-cpu-baseline --no-others --sustain-s 0 --steps 100 --warmup 20 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2; do
  echo "product $(run)"
  for st in 0 1 2 3; do echo "persistent stagger $st $(MOF_PC_STAGGER=$st MOF_LIB_PATH=/tmp/libmof_p64.so run)"; done
done
