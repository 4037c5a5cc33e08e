#!/bin/bash
# Diagnostic: rebuild sr_kernel.o with ablation macros on the GPU box and time c5 (results are WRONG by design).
cd ${GRAFT_REPO_ROOT:-/root/repo}/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
for V in "$@"; do
  hipcc --offload-arch=gfx950 $BASE $V -I../../include -I. -c -o sr_kernel.o sr_kernel.hip 2>/dev/null && \
  hipcc --offload-arch=gfx950 -shared -o ../libmof_hip.so mof_capi.o mof_sr.o pc_kernel.o pc_kernel_quad.o pc_kernel_mixed.o bm_kernel.o sr_kernel.o && \
  echo "variant [$V]: $(python3 ../../bench.py --workload c5 --no-cpu-baseline | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
