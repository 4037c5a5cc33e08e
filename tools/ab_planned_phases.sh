#!/bin/bash
# Where the planned kernel's time goes (p60 by default): phase ablations of pc_kernel_generic.hip, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-p60}
bash $R/tools/ab_variants.sh pc_kernel_generic.hip "--workload $WL --steps 50 --warmup 10" "" "-DMOF_GABL=1" "-DMOF_GABL=2" "-DMOF_GABL=3" "-DMOF_GABL=4" 2>/dev/null | grep "rep [12]"
