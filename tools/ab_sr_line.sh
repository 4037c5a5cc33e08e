#!/bin/bash
# LDS line pitch of the estimator's 480-point kernels (SrPlan<480>::LINE, sr_common.hpp): per-kernel times for a list of pitches, same box.
# The three translation units that include sr_common.hpp are rebuilt per value into /tmp; the product library is not touched.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
OTHERS=$(ls *.hip | grep -v "^sr_kernel.hip$\|^sr_seq_kernel.hip$\|^sr_fused_kernel.hip$\|^pc_kernel_quad.hip$" | sed 's/\.hip$/.o/')
i=0
for L in "$@"; do
  for f in sr_kernel sr_seq_kernel sr_fused_kernel; do hipcc --offload-arch=gfx950 $BASE -DMOF_SR_LINE480=$L -I../../include -I. -c -o /tmp/line_${f}.o $f.hip || exit 1; done
  hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_ab_$i.so $OTHERS /tmp/line_sr_kernel.o /tmp/line_sr_seq_kernel.o /tmp/line_sr_fused_kernel.o -ldl
  i=$((i+1))
done
bash $R/tools/ab_stats.sh "--workload c5 --steps 10 --warmup 3" $i 2>&1 | grep "variant\|rows_real\|cols_seq\|rows_inv"
