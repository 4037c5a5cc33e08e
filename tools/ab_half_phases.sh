#!/bin/bash
# Phase ablation of the half-tile kernel (csrc/pc_half_kernel.hip, -DMOF_HABL=k: results wrong by design): 0 product, 1 no transform
# passes, 2 no cross-power, 3 no pixel loads. usage (on the GPU box): bash tools/ab_half_phases.sh <workload> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=$1; shift
export MOF_FFT_HALF=1
bash $R/tools/ab_variants.sh pc_half_kernel.hip "--workload $WL --steps 30 --warmup 10 $*" "-DMOF_HABL=0" "-DMOF_HABL=1" "-DMOF_HABL=2" "-DMOF_HABL=3" 2>&1 | grep -v amdgpu.ids
