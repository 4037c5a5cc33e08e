#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export MOF_SR_FUSED=1
bash $R/tools/ab_variants.sh sr_fused_kernel.hip "--workload c5 --steps 10 --warmup 3" "" "-DMOF_FUSED_UNROLL=3" "-DMOF_FUSED_UNROLL=5" > /dev/null 2>&1
bash $R/tools/ab_stats.sh "--workload c5 --steps 10 --warmup 3" 3 2>&1 | grep "variant\|fused"
