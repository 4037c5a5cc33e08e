#!/bin/bash
# Same-box sweep of the half-tile kernel's line pitch / skew at ONE transform size (csrc/pc_half_kernel.hip: -DMOF_HALF_ONLY=<m>
# -DMOF_HALF_PITCH=<p> -DMOF_HALF_SKEW=<0|1>): the bank model (tools/design/half_banks.py) proposes, the box disposes.
#   usage (on the GPU box): bash tools/sweep_half_pitch.sh <m> <workload> "<p:skew[:shift]> <p:skew[:shift]> ..."   (shift: the skew's
#   shift, -DMOF_HALF_SHIFT, 3 when omitted)
R=${GRAFT_REPO_ROOT:-/root/repo}
M=$1; WL=$2; shift 2
export MOF_FFT_HALF=1
V=()
for ps in $1; do
  IFS=: read -r pp sk sh <<< "$ps"
  V+=("-DMOF_HALF_ONLY=$M -DMOF_HALF_PITCH=$pp -DMOF_HALF_SKEW=$sk -DMOF_HALF_SHIFT=${sh:-3}")
done
echo "variants: $1"
bash $R/tools/ab_variants.sh pc_half_kernel.hip "--workload $WL --steps 40 --warmup 10" "${V[@]}" 2>&1 | grep -v amdgpu.ids
