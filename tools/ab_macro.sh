#!/bin/bash
# A/B on ONE box (boxes differ by ~5 % in clock): pc_kernel.hip + pc_kernel_mixed.hip built with two macro sets ($1 vs $2), c2 x3 each
cd ${GRAFT_REPO_ROOT:-/root/repo}/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
cp ../libmof_hip.so /tmp/lib_keep.so
i=0
for V in "$1" "$2"; do
  hipcc --offload-arch=gfx950 $BASE $V -I../../include -I. -c -o /tmp/pc_$i.o pc_kernel.hip
  hipcc --offload-arch=gfx950 $BASE $V -I../../include -I. -c -o /tmp/pcm_$i.o pc_kernel_mixed.hip
  hipcc --offload-arch=gfx950 -shared -o /tmp/lib_$i.so mof_capi.o mof_sr.o /tmp/pc_$i.o pc_kernel_quad.o /tmp/pcm_$i.o bm_kernel.o sr_kernel.o
  i=$((i+1))
done
for rep in 1 2 3; do
  for v in 0 1; do
    cp /tmp/lib_$v.so ../libmof_hip.so
    echo "variant$v $(python3 ../../bench.py --no-cpu-baseline --steps 50 ${AB_ARGS} | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["roofline"]["kernel_ms"])')"
  done
done
cp /tmp/lib_keep.so ../libmof_hip.so
