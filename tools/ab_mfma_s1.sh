#!/bin/bash
# K1 (N = 64) with S1 of the forward transform on the matrix cores (-DMOF_K1_MFMA_S1=1, pc_passes3.hpp: fwd3_rows_mfma) against the
# product kernel: same-box interleaved rates (tools/ab_variants.sh), then the SQ / TCP counters of both builds (separate --pmc passes).
#   v0 product   v1 MFMA S1, accumulators in AGPRs (compiler default)   v2 MFMA S1, accumulators in VGPRs (-amdgpu-mfma-vgpr-form)
# usage (GPU box): bash tools/ab_mfma_s1.sh   -> gpurun_out/r04_mfma_s1_ab.txt, gpurun_out/r04_mfma_s1_{v0,v2}_pmc.csv
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab_variants.sh pc_kernel.hip "--workload c2" "" "-DMOF_K1_MFMA_S1=1" "-DMOF_K1_MFMA_S1=1 -mllvm -amdgpu-mfma-vgpr-form" 2>/dev/null > $R/gpurun_out/r04_mfma_s1_ab.txt
G1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
G3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
for v in 0 2; do
  MOF_LIB_PATH=/tmp/libmof_ab_$v.so bash $R/tools/pmc.sh r04_mfma_s1_v$v "$G1" "$G2" "$G3" -- --workload c2 --no-others --sustain-s 0 --steps 20 --warmup 5
  python3 $R/tools/pmc_table.py $R/gpurun_out/prof_r04_mfma_s1_v$v > $R/gpurun_out/r04_mfma_s1_v${v}_pmc.csv
done
cat $R/gpurun_out/r04_mfma_s1_ab.txt
