#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export MOF_SR_COLS_SPLIT=1
bash $R/tools/ab_variants.sh sr_seq_kernel.hip "--workload c5 --steps 10 --warmup 3" "-DMOF_K6P_WPE=2" "-DMOF_K6P_WPE=3" "-DMOF_K6P_WPE=4" > /dev/null 2>&1
bash $R/tools/ab_stats.sh "--workload c5 --steps 10 --warmup 3" 3 2>&1 | grep "variant\|cols_"
