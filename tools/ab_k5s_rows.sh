#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab_variants.sh sr_seq_kernel.hip "--workload c5 --steps 10 --warmup 3" "-DMOF_K5S_ROWS32=1" "-DMOF_K5S_ROWS32=0" > /dev/null 2>&1
bash $R/tools/ab_stats.sh "--workload c5 --steps 10 --warmup 3" 2 2>&1 | grep "variant\|rows_real"
