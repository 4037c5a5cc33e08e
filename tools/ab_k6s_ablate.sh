#!/bin/bash
# Is K6s (sr_cols_seq_kernel) bound by its HBM reads or by its own work? Same box: the product, and every wave reading pair 0's lines (L2 hits)
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab_variants.sh sr_seq_kernel.hip "--workload c5 --steps 10 --warmup 3" "" "-DMOF_K6S_ABLATE=1" > /dev/null 2>&1
bash $R/tools/ab_stats.sh "--workload c5 --steps 10 --warmup 3" 2 2>&1 | grep "variant\|cols_seq\|rows_real\|rows_inv"
